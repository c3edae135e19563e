"""HIP path vs the CPU oracle on identical inputs, through the C ABI (bit-exact: the kernels follow the
oracle's arithmetic contract, oracle/lexlse_oracle.h).  Tolerance north_star asks for: pivots/ranks
exact, x within 1e-10; what is asserted here is stronger (bitwise equality) wherever noted."""
import numpy as np
import pytest

from lexls_amd import problems as P

pytestmark = pytest.mark.gpu


def run_both(hip, oracle, lod, dims, nvar, maxdim=None, keep_factor=True, force_generic=False, **fixed):
    batch = lod.shape[0]
    dims_a = np.asarray(dims, np.uint32)
    if maxdim is None:
        maxdim = dims_a if dims_a.ndim == 1 else dims_a.max(axis=0)
        maxdim = np.array(maxdim, np.uint32)
        maxdim[-1] += lod.shape[2] - int(maxdim.sum())
    ref = oracle.lse_run(lod, dims, nvar, maxdim=maxdim, **fixed)
    s = hip.BatchedLexLSE(batch, nvar, maxdim)
    s.set_kernel_policy(force_generic)
    s.setObjDim(dims_a)
    if "nfixed" in fixed:
        s.fixVariables(fixed["nfixed"], fixed["fixed_idx"], fixed["fixed_val"], fixed.get("fixed_type"))
    s.setProblem(lod)
    s.factorize_solve(keep_factor=keep_factor)
    return s, ref


def assert_factor_equal(s, ref, dims, nvar):
    r, fc, tr = s.getRanks()
    np.testing.assert_array_equal(r, ref["rank"])
    np.testing.assert_array_equal(fc, ref["fcol"])
    np.testing.assert_array_equal(tr, ref["totalrank"])
    np.testing.assert_array_equal(s.get_column_permutations(), ref["perm"])
    np.testing.assert_array_equal(s.get_hh_scalars(), ref["hh"])
    f = s.get_lexqr()
    dims_a = np.asarray(dims)
    m = dims_a.sum(axis=-1) if dims_a.ndim == 2 else np.full(f.shape[0], dims_a.sum())
    for b in range(f.shape[0]):
        np.testing.assert_array_equal(f[b, :, :m[b]], ref["factor"][b, :, :m[b]])


def assert_factor_close(s, ref, dims, nvar, tol=1e-10):
    """north_star's contract for the fast large path: ranks, first columns and pivots exact, values within 1e-10 RELATIVE to the largest
    entry of the quantity compared (never below 1): a factor whose entries grow with the problem's scale is held to the same number of digits"""
    r, fc, tr = s.getRanks()
    np.testing.assert_array_equal(r, ref["rank"])
    np.testing.assert_array_equal(fc, ref["fcol"])
    np.testing.assert_array_equal(tr, ref["totalrank"])
    np.testing.assert_array_equal(s.get_column_permutations(), ref["perm"])
    assert np.abs(s.get_hh_scalars() - ref["hh"]).max() <= tol * max(1.0, float(np.abs(ref["hh"]).max()))
    f = s.get_lexqr()
    dims_a = np.asarray(dims)
    m = dims_a.sum(axis=-1) if dims_a.ndim == 2 else np.full(f.shape[0], dims_a.sum())
    for b in range(f.shape[0]):
        scale = max(1.0, float(np.abs(ref["factor"][b, :, :m[b]]).max()))
        assert np.abs(f[b, :, :m[b]] - ref["factor"][b, :, :m[b]]).max() <= tol * scale
    assert np.abs(s.get_x() - ref["x"]).max() <= tol * max(1.0, float(np.abs(ref["x"]).max()))


LARGE_PATHS = pytest.mark.parametrize("policy", [0, 5], ids=["step-per-pivot", "bit-exact-multi-launch"])


def check_large(s, ref, dims, n, policy):
    if policy == 5:
        assert s.last_kernel() == "lqr_large<multi-launch>"
        assert_factor_equal(s, ref, dims, n)
        np.testing.assert_array_equal(s.get_x(), ref["x"])
    else:
        assert s.last_kernel() == "lqr_large<step-per-pivot,mfma>"
        assert_factor_close(s, ref, dims, n)


# kernel policy (include/lexls_hip.h): 0 = automatic dispatch (small shapes: register-resident wave kernel up to one round of it, left-looking
# beyond), 1 = generic kernel only, 2 = never the left-looking wave kernel, 3 = the left-looking wave kernel whenever the shape allows
# 4 = the four-problems-per-wavefront kernel (x-only solves of shapes it serves; anything else falls through to the automatic choice)
BOTH_PATHS = pytest.mark.parametrize("force_generic", [0, 1, 2, 3, 4], ids=["automatic", "generic", "register-resident", "left-looking", "quad"])


@BOTH_PATHS
def test_ik_batch_bit_exact(hip, oracle, force_generic):
    dims, n = [12] * 5, 40
    lod = P.lse_batch(20260100, 64, n, dims)
    s, ref = run_both(hip, oracle, lod, dims, n, force_generic=force_generic)
    assert (ref["rank"] == [12, 12, 12, 4, 0]).all()
    assert_factor_equal(s, ref, dims, n)
    np.testing.assert_array_equal(s.get_x(), ref["x"])
    assert s.last_kernel() == {0: "lqr_wave<41,12,exact>", 1: "lqr_generic<64,lds>", 2: "lqr_wave<41,12,exact>", 3: "lqr_lwave<41,12,exact>", 4: "lqr_quad<3,12,shift 7,factor>"}[force_generic]


@BOTH_PATHS
def test_x_only_variant_matches(hip, oracle, force_generic):
    dims, n = [12] * 5, 40
    lod = P.lse_batch(7, 32, n, dims)
    s, ref = run_both(hip, oracle, lod, dims, n, keep_factor=False, force_generic=force_generic)
    if force_generic == 0:
        # automatic dispatch: the tolerance-contract kernel (north_star: pivots / ranks exact, x within 1e-10; tests/test_gpu_qtol.py)
        assert s.last_kernel() == "lqr_qtol<3,12,shift 7>"
        assert np.abs(s.get_x() - ref["x"]).max() <= 1e-10
    else:
        np.testing.assert_array_equal(s.get_x(), ref["x"])
    np.testing.assert_array_equal(s.getRanks()[0], ref["rank"])
    np.testing.assert_array_equal(s.get_column_permutations(), ref["perm"])
    if force_generic == 4:
        assert s.last_kernel() == "lqr_quad<3,12,shift 7>"  # n = 40: the right-aligned layout
    with pytest.raises(Exception):
        s.get_lexqr()  # factor was not kept: the library must refuse, not return stale data


def test_factorize_then_solve_separately(hip, oracle):
    dims, n = [6] * 5, 40
    lod = P.lse_batch(11, 16, n, dims)
    ref = oracle.lse_run(lod, dims, n)
    s = hip.BatchedLexLSE(16, n, dims)
    s.setProblem(lod)
    s.factorize()
    s.solve()
    np.testing.assert_array_equal(s.get_x(), ref["x"])
    assert (ref["rank"] == 6).all()


@BOTH_PATHS
def test_rank_deficient_levels(hip, oracle, force_generic):
    n, dims, ranks = 15, [5, 5, 5, 5], [3, 3, 3, 3]
    lod = np.stack([P.rank_deficient_problem(100 + b, n, dims, ranks) for b in range(24)])
    s, ref = run_both(hip, oracle, lod, dims, n, force_generic=force_generic)
    assert (ref["rank"] == ranks).all()
    assert_factor_equal(s, ref, dims, n)
    np.testing.assert_array_equal(s.get_x(), ref["x"])


@BOTH_PATHS
def test_ragged_batch(hip, oracle, force_generic):
    n, cap_dims = 20, [8, 8, 8]
    rng_dims = np.array([[8, 8, 8], [3, 0, 5], [1, 8, 2], [0, 0, 4], [8, 1, 0], [5, 5, 5], [2, 2, 2], [7, 3, 8]], np.uint32)
    full = np.zeros((8, n + 1, 24))
    for b in range(8):
        m = int(rng_dims[b].sum())
        full[b, :, :m] = P.lse_problem(900 + b, n, rng_dims[b])
    s, ref = run_both(hip, oracle, full, rng_dims, n, maxdim=np.array(cap_dims, np.uint32), force_generic=force_generic)
    assert_factor_equal(s, ref, rng_dims, n)
    np.testing.assert_array_equal(s.get_x(), ref["x"])




@pytest.mark.parametrize("n,dims", [(63, [16, 16, 16, 16]), (30, [14, 9, 16]), (40, [6] * 5), (5, [12, 12]), (40, [12, 0, 12, 12, 12]), (12, [1, 1, 1, 1, 1, 1, 1, 1])])
@pytest.mark.parametrize("policy", [3, 2], ids=["left-looking-if-it-fits", "register-resident"])
def test_wave_kernel_shapes(hip, oracle, n, dims, policy):
    lod = P.lse_batch(1000 + n, 9, n, dims)
    s, ref = run_both(hip, oracle, lod, dims, n, force_generic=policy)
    assert "wave<" in s.last_kernel()
    if policy == 3 and n <= 40 and max(dims) <= 12:
        assert s.last_kernel().startswith("lqr_lwave")
    assert_factor_equal(s, ref, dims, n)
    np.testing.assert_array_equal(s.get_x(), ref["x"])


def _quad_name(n, factor=False):
    """the four-per-wavefront instantiation level dims <= 12 dispatch to: one / two / three slots of sixteen columns (n = 40: right-aligned)"""
    slots = 1 if n + 1 <= 16 else (2 if n + 1 <= 32 else 3)
    return f"lqr_quad<{slots},12{',shift 7' if n == 40 else ''}{',factor' if factor else ''}>"


def _quad_x_only(hip, oracle, lod, dims, n, maxdim=None, kernel="lqr_quad<3,12>"):
    """x-only solve on the four-problems-per-wavefront kernel: x, ranks, first columns and pivots bit for bit against the oracle"""
    s, ref = run_both(hip, oracle, lod, dims, n, maxdim=maxdim, keep_factor=False, force_generic=4)
    assert s.last_kernel() == (_quad_name(n) if kernel == "lqr_quad<3,12>" else kernel)
    r, fc, tr = s.getRanks()
    np.testing.assert_array_equal(r, ref["rank"])
    np.testing.assert_array_equal(fc, ref["fcol"])
    np.testing.assert_array_equal(tr, ref["totalrank"])
    np.testing.assert_array_equal(s.get_column_permutations(), ref["perm"])
    np.testing.assert_array_equal(s.get_x(), ref["x"])
    return s, ref


@pytest.mark.parametrize("n,dims", [(47, [12, 12, 12, 12]), (30, [9, 12, 5]), (40, [6] * 5), (5, [12, 12]), (40, [12, 0, 12, 12, 12]), (12, [1] * 8),
                                    (33, [11, 7, 12, 3]), (16, [12, 12]), (15, [8, 8]), (32, [12, 12, 12])])
@pytest.mark.parametrize("batch", [13, 4, 1])
def test_quad_kernel_shapes(hip, oracle, n, dims, batch):
    """shapes around the 16-column slot boundaries, batches that leave rows of the last wavefront idle"""
    _quad_x_only(hip, oracle, P.lse_batch(2000 + n, batch, n, dims), dims, n)


@pytest.mark.parametrize("n,dims", [(63, [16, 16, 16, 16]), (30, [14, 9, 16]), (48, [12, 12, 12, 12]), (55, [16, 8, 16, 16]), (40, [16] * 3), (20, [15, 13])])
def test_quad_kernel_wide_shapes(hip, oracle, n, dims):
    """the <4 slots, 16 rows> instantiation: n + 1 <= 64, level dims <= 16 (the IK families beyond 47 variables / 12 rows per level)"""
    _quad_x_only(hip, oracle, P.lse_batch(3000 + n, 7, n, dims), dims, n, kernel="lqr_quad<4,16>")


@pytest.mark.parametrize("n,dims", [(47, [12, 12, 12, 12]), (30, [9, 12, 5]), (40, [6] * 5), (5, [12, 12]), (40, [12, 0, 12, 12, 12]), (12, [1] * 8), (33, [11, 7, 12, 3]),
                                    (40, [12] * 5)])
def test_quad_kernel_factor_output(hip, oracle, n, dims):
    """the factor-keeping instantiation: get_lexqr layout (multipliers, R / T blocks, essential parts, eliminated rows of the levels after the
    columns ran out), Householder scalars, pivots and x bit for bit"""
    s, ref = run_both(hip, oracle, P.lse_batch(4000 + n, 11, n, dims), dims, n, keep_factor=True, force_generic=4)
    assert s.last_kernel() == _quad_name(n, factor=True)
    assert_factor_equal(s, ref, dims, n)
    np.testing.assert_array_equal(s.get_x(), ref["x"])
    np.testing.assert_array_equal(s.get_v(), ref["v"])  # the residual kernel reads the stored factor


def test_quad_kernel_factor_output_rank_deficient_and_ragged(hip, oracle):
    """dependent rows of a rank-deficient level wait at their current position and move with later swaps; wavefronts whose rows stop at
    different pivots keep the stopped rows' blocks"""
    n, dims = 15, [5, 5, 5, 5]
    lod = np.stack([P.rank_deficient_problem(500 + b, n, dims, [3, 3, 3, 3] if b % 2 else [5, 2, 4, 1]) for b in range(14)])
    s, ref = run_both(hip, oracle, lod, dims, n, keep_factor=True, force_generic=4)
    assert s.last_kernel() == _quad_name(n, factor=True)
    assert_factor_equal(s, ref, dims, n)
    np.testing.assert_array_equal(s.get_x(), ref["x"])
    n3, cap3 = 20, [8, 8, 8]
    rd = np.array([[8, 8, 8], [3, 0, 5], [1, 8, 2], [0, 0, 4], [8, 1, 0], [5, 5, 5], [2, 2, 2], [7, 3, 8]], np.uint32)
    full = np.zeros((8, n3 + 1, 24))
    for b in range(8):
        m = int(rd[b].sum())
        full[b, :, :m] = P.lse_problem(900 + b, n3, rd[b])
    s, ref = run_both(hip, oracle, full, rd, n3, maxdim=np.array(cap3, np.uint32), keep_factor=True, force_generic=4)
    assert_factor_equal(s, ref, rd, n3)
    np.testing.assert_array_equal(s.get_x(), ref["x"])


def test_quad_kernel_rank_deficient_and_ragged(hip, oracle):
    """rows of one wavefront stop at different pivots / have different level sizes: the masked elimination path and the frozen bookkeeping"""
    n, dims, ranks = 15, [5, 5, 5, 5], [3, 3, 3, 3]
    lod = np.stack([P.rank_deficient_problem(100 + b, n, dims, ranks) for b in range(24)])
    _, ref = _quad_x_only(hip, oracle, lod, dims, n)
    assert (ref["rank"] == ranks).all()
    # a wavefront that mixes full-rank and rank-deficient problems
    mixed = np.stack([P.rank_deficient_problem(300 + b, n, dims, [5, 5, 5, 0] if b % 3 else [2, 4, 1, 5]) for b in range(10)])
    _quad_x_only(hip, oracle, mixed, dims, n)
    n3, cap3 = 20, [8, 8, 8]
    rd = np.array([[8, 8, 8], [3, 0, 5], [1, 8, 2], [0, 0, 4], [8, 1, 0], [5, 5, 5], [2, 2, 2], [7, 3, 8]], np.uint32)
    full = np.zeros((8, n3 + 1, 24))
    for b in range(8):
        m = int(rd[b].sum())
        full[b, :, :m] = P.lse_problem(900 + b, n3, rd[b])
    _quad_x_only(hip, oracle, full, rd, n3, maxdim=np.array(cap3, np.uint32))


def test_quad_kernel_tied_norms_and_degenerate_reflectors(hip, oracle):
    """exact ties resolve to the first position; columns with a single non-zero give H = I (tau = 0) in some rows of a wavefront only"""
    n, dims = 10, [4, 4, 4]
    lod = P.lse_batch(77, 7, n, dims)
    lod[:, 3, :] = lod[:, 1, :]
    lod[:, 7, :] = lod[:, 1, :]
    lod[:, 5, :] = -lod[:, 2, :]
    _quad_x_only(hip, oracle, lod, dims, n)
    n2, dims2 = 12, [6, 6]
    lod2 = P.lse_batch(78, 9, n2, dims2)
    lod2[::2, :n2, :6] = 0.0
    for j in range(6):
        lod2[::2, 2 * j, j] = 3.0 + j  # identity-like level in every other problem: degenerate reflectors next to regular ones
    _quad_x_only(hip, oracle, lod2, dims2, n2)


@pytest.mark.parametrize("policy", [3, 2], ids=["left-looking", "register-resident"])
def test_wave_kernel_tied_norms(hip, oracle, policy):
    """duplicated columns: exact ties in the pivot search must resolve to the first position (maxCoeff semantics)."""
    n, dims = 10, [4, 4, 4]
    lod = P.lse_batch(77, 4, n, dims)
    lod[:, 3, :] = lod[:, 1, :]
    lod[:, 7, :] = lod[:, 1, :]
    lod[:, 5, :] = -lod[:, 2, :]
    s, ref = run_both(hip, oracle, lod, dims, n, force_generic=policy)
    assert_factor_equal(s, ref, dims, n)
    np.testing.assert_array_equal(s.get_x(), ref["x"])


@BOTH_PATHS
def test_fixed_variables(hip, oracle, force_generic):
    n, dims, batch = 12, [4, 4, 6], 10
    lod = P.lse_batch(31, batch, n, dims)
    nfixed = np.array([0, 1, 2, 3, 4, 5, 12, 2, 1, 3], np.uint32)
    idx = np.zeros((batch, n), np.uint32)
    val = np.zeros((batch, n))
    typ = np.full((batch, n), 2, np.uint8)
    for b in range(batch):
        perm = np.argsort(P.uniform(500 + b, n))
        idx[b, :nfixed[b]] = perm[:nfixed[b]]
        val[b, :nfixed[b]] = P.normal(600 + b, n)[:nfixed[b]]
    s, ref = run_both(hip, oracle, lod, dims, n, force_generic=force_generic, nfixed=nfixed, fixed_idx=idx, fixed_val=val, fixed_type=typ)
    assert_factor_equal(s, ref, dims, n)
    np.testing.assert_array_equal(s.get_x(), ref["x"])
    assert s.last_kernel().startswith({1: "lqr_generic", 4: "lqr_quad<3,12,factor,fixed>"}.get(force_generic, "lqr_wave"))
    if force_generic in (0, 4):  # x only: automatic dispatch and policy 4 take the four-per-wavefront kernel's fixed-variable form
        s.factorize_solve(keep_factor=False)
        assert s.last_kernel() == "lqr_quad<3,12,fixed>"
        np.testing.assert_array_equal(s.get_x(), ref["x"])
        np.testing.assert_array_equal(s.get_column_permutations(), ref["perm"])
        np.testing.assert_array_equal(s.getRanks()[0], ref["rank"])


@pytest.mark.parametrize("keep", [False, True], ids=["x-only", "factor"])
def test_quad_kernel_fixed_variables_ik_shape(hip, oracle, keep):
    """the four-per-wavefront kernel with fixed variables (FIX instantiations) on the IK shape (right-aligned layout), ragged numbers of fixed
    variables inside a wavefront (0 .. 9 and all 40), chained indices, rank-deficient problems: bit-identical to the oracle"""
    n, dims, batch = 40, [12] * 5, 23
    lod = np.concatenate([P.lse_batch(3300, batch - 6, n, dims), np.stack([P.rank_deficient_problem(3400 + b, n, dims, [7, 12, 3, 9, 2]) for b in range(6)])])
    nfixed = np.array([0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 40, 3, 1, 2, 12, 5, 0, 4, 6, 2, 3, 1, 7], np.uint32)
    idx = np.zeros((batch, n), np.uint32)
    val = np.zeros((batch, n))
    for b in range(batch):
        perm = np.argsort(P.uniform(3500 + b, n))
        idx[b, :nfixed[b]] = perm[:nfixed[b]]
        val[b, :nfixed[b]] = P.normal(3600 + b, n)[:nfixed[b]]
    idx[11, :3] = [5, 0, 1]  # chained: position 0 is taken by the first swap
    s, ref = run_both(hip, oracle, lod, dims, n, keep_factor=keep, force_generic=4, nfixed=nfixed, fixed_idx=idx, fixed_val=val)
    assert s.last_kernel() == ("lqr_quad<3,12,shift 7,factor,fixed>" if keep else "lqr_quad<3,12,shift 7,fixed>")
    np.testing.assert_array_equal(s.get_x(), ref["x"])
    np.testing.assert_array_equal(s.get_column_permutations(), ref["perm"])
    np.testing.assert_array_equal(s.getRanks()[0], ref["rank"])
    np.testing.assert_array_equal(s.getRanks()[1], ref["fcol"])
    if keep:
        assert_factor_equal(s, ref, dims, n)


def test_fixed_variables_chained_indices(hip, oracle):
    """fixed indices that point at positions already used by earlier swaps exercise the chained index rule (lexlse.h:146-153)"""
    n, dims, batch = 8, [3, 4], 6
    lod = P.lse_batch(131, batch, n, dims)
    idx = np.zeros((batch, n), np.uint32)
    val = np.zeros((batch, n))
    pats = [[3, 0, 1], [1, 0, 2], [2, 1, 0], [7, 0, 1], [0, 1, 2], [5, 0, 5 - 5 + 3]]
    for b in range(batch):
        idx[b, :3] = pats[b]
        val[b, :3] = P.normal(140 + b, 3)
    nfixed = np.full(batch, 3, np.uint32)
    for fg in (False, True):
        s, ref = run_both(hip, oracle, lod, dims, n, force_generic=fg, nfixed=nfixed, fixed_idx=idx, fixed_val=val)
        assert_factor_equal(s, ref, dims, n)
        np.testing.assert_array_equal(s.get_x(), ref["x"])


def test_medium_problem_256_thread_variant(hip, oracle):
    n, dims = 100, [30, 30, 30, 30]
    lod = P.lse_batch(41, 3, n, dims)
    s, ref = run_both(hip, oracle, lod, dims, n)
    assert_factor_equal(s, ref, dims, n)
    np.testing.assert_array_equal(s.get_x(), ref["x"])
    assert s.last_kernel() == "lqr_generic<256,lds>"


@pytest.mark.parametrize("force_generic", [0, 1, 5], ids=["step-per-pivot", "generic", "bit-exact-multi-launch"])
def test_hbm_resident_variants(hip, oracle, force_generic):
    """problems too large for a CU's LDS: the two large paths, and the one-workgroup generic fallback"""
    n, dims = 200, [100, 100, 100, 100]
    lod = P.lse_batch(43, 2, n, dims)
    s, ref = run_both(hip, oracle, lod, dims, n, force_generic=force_generic)
    if force_generic == 1:
        assert s.last_kernel() == "lqr_generic<1024,hbm>"
        assert_factor_equal(s, ref, dims, n)
        np.testing.assert_array_equal(s.get_x(), ref["x"])
    else:
        check_large(s, ref, dims, n, force_generic)


@LARGE_PATHS
def test_config2_single_large(hip, oracle, policy):
    """BASELINE.json configs[1]: n=512, 4 levels x 256 rows (large path): pivots / ranks exact and values within 1e-10 on the step-per-pivot
    path (tree sums, trailing update on the matrix cores), bit for bit on the ordered-chain path."""
    n, dims = 512, [256] * 4
    lod = P.lse_batch(20260001, 1, n, dims)
    s, ref = run_both(hip, oracle, lod, dims, n, force_generic=policy)
    assert (ref["rank"] == [256, 256, 0, 0]).all()
    check_large(s, ref, dims, n, policy)
    # per-level residuals (get_v, lexlse.h:1560-1582) within the same tolerance, relative to the largest residual entry
    assert np.abs(s.get_v() - ref["v"]).max() <= (0.0 if policy == 5 else 1e-10 * max(1.0, float(np.abs(ref["v"]).max())))


@LARGE_PATHS
def test_large_path_rank_deficient_and_ragged(hip, oracle, policy):
    n, cap_dims = 150, [90, 90, 90]
    rdims = np.array([[90, 90, 90], [60, 0, 85], [90, 40, 7]], np.uint32)
    full = np.zeros((3, n + 1, 270))
    full[0, :, :270] = P.rank_deficient_problem(801, n, [90, 90, 90], [50, 40, 30])
    for b in (1, 2):
        m = int(rdims[b].sum())
        full[b, :, :m] = P.lse_problem(802 + b, n, rdims[b])
    s, ref = run_both(hip, oracle, full, rdims, n, maxdim=np.array(cap_dims, np.uint32), force_generic=policy)
    assert ref["rank"][0].tolist() == [50, 40, 30]
    check_large(s, ref, rdims, n, policy)
    assert np.abs(s.get_v() - ref["v"]).max() <= (0.0 if policy == 5 else 1e-10)


@pytest.mark.parametrize("mode", ["1", "0", "2"])
def test_large_path_one_launch_per_level(hip, oracle, monkeypatch, mode):
    """single large problems run the pivots of a level inside ONE launch (default; LEXLS_LARGE_PERSIST=1): workgroups hand their candidate
    records and the pivot column to each other as tagged 16-byte granules (agent-scope sc1 stores / loads, every spin bounded).  Same
    contract as the launch per pivot (LEXLS_LARGE_PERSIST=0), also when the launch gives up and the level is redone pivot by pivot
    (LEXLS_LARGE_PERSIST=2 raises the abort flag before the launch), and on a rank-deficient problem whose levels stop early"""
    monkeypatch.setenv("LEXLS_LARGE_PERSIST", mode)
    n, dims = 150, [90, 90, 90]
    lod = P.rank_deficient_problem(811, n, dims, [60, 50, 30])[None]
    s, ref = run_both(hip, oracle, lod, dims, n)
    assert ref["rank"][0].tolist() == [60, 50, 30]
    check_large(s, ref, dims, n, 0)
    lod = P.lse_batch(5, 1, n, dims)  # full rank: the columns run out inside level 1
    s, ref = run_both(hip, oracle, lod, dims, n)
    check_large(s, ref, dims, n, 0)


@LARGE_PATHS
def test_large_path_ragged_rows_below(hip, oracle, policy):
    """a problem that is NOT the largest of the batch has the most rows below level 0 (the Gauss step's row grid must cover it)"""
    n, cap_dims = 150, [200, 200]
    rdims = np.array([[200, 10], [10, 200], [50, 150]], np.uint32)
    full = np.zeros((3, n + 1, 400))
    for b in range(3):
        m = int(rdims[b].sum())
        full[b, :, :m] = P.lse_problem(880 + b, n, rdims[b])
    s, ref = run_both(hip, oracle, full, rdims, n, maxdim=np.array(cap_dims, np.uint32), force_generic=policy)
    check_large(s, ref, rdims, n, policy)


def test_residuals(hip, oracle):
    n, dims = 15, [5, 5, 5, 5]
    lod = np.stack([P.rank_deficient_problem(200 + b, n, dims, [3, 3, 3, 3]) for b in range(8)])
    s, ref = run_both(hip, oracle, lod, dims, n)
    np.testing.assert_array_equal(s.get_v(), ref["v"])
    # and they are the true residuals A x - b
    x = s.get_x()
    for b in range(8):
        A = lod[b, :-1, :].T
        np.testing.assert_allclose(A @ x[b] - lod[b, -1, :], ref["v"][b], atol=1e-10)


@pytest.mark.parametrize("level", [0, 1, 2, 3])
def test_objective_sensitivity(hip, oracle, level):
    n, dims, batch = 15, [5, 5, 5, 5], 12
    lod = np.stack([P.rank_deficient_problem(300 + b, n, dims, [3, 4, 3, 2]) for b in range(batch)])
    types = np.zeros((batch, 20), np.uint8)
    for b in range(batch):
        types[b] = 1 + (P.uniform(700 + b, 20) * 3).astype(np.uint8)  # LB / UB / EQ
    ref = oracle.lse_run(lod, dims, n, ctr_type=types, sens_obj=level)
    s = hip.BatchedLexLSE(batch, n, dims)
    s.setProblem(lod)
    s.setCtrType(types)
    s.factorize_solve()
    found, ctr, obj, maxabs = s.ObjectiveSensitivity(level)
    np.testing.assert_array_equal(found.astype(np.int32), ref["sens"][:, 0])
    np.testing.assert_array_equal(ctr, ref["sens"][:, 1])
    np.testing.assert_array_equal(obj, ref["sens"][:, 2])
    np.testing.assert_array_equal(maxabs, ref["maxabs"])
    np.testing.assert_array_equal(s.getWorkspace(), ref["lam"])
    np.testing.assert_array_equal(s.getCtrType(), ref["ctr_type_out"])


@pytest.mark.parametrize("start", [0, 1])
def test_sensitivity_scan_equals_level_by_level_calls(hip, oracle, start):
    """lexls_lse_set_sensitivity_scan: one launch does what LexLSI's removal search does with one ObjectiveSensitivity call per level
    (lexlsi.h:1121-1132) — stop at the first level that reports a wrong-sign multiplier, carry the CORRECT_SIGN marks along."""
    n, dims, batch = 15, [5, 5, 5, 5], 16
    lod = np.stack([P.rank_deficient_problem(900 + b, n, dims, [3, 4, 3, 2]) for b in range(batch)])
    types = np.zeros((batch, 20), np.uint8)
    for b in range(batch):
        types[b] = 1 + (P.uniform(950 + b, 20) * 3).astype(np.uint8)  # LB / UB / EQ
        if b % 3 == 0:
            types[b, :10] = 3  # equalities on the first levels: the search has to go further up
    # reference: the level loop of the driver, problem by problem (each one stops at its own level)
    sens = np.zeros((batch, 3), np.int32)
    maxabs, lam, marks = np.zeros(batch), np.zeros((batch, n + 20)), types.copy()
    stopped = np.zeros(batch, int)
    for b in range(batch):
        cur = types[b:b + 1].copy()
        for level in range(start, len(dims)):
            ref = oracle.lse_run(lod[b:b + 1], dims, n, ctr_type=cur, sens_obj=level)
            cur = ref["ctr_type_out"]
            if ref["sens"][0, 0] or level == len(dims) - 1:
                sens[b], maxabs[b], lam[b], marks[b], stopped[b] = ref["sens"][0], ref["maxabs"][0], ref["lam"][0], cur[0], level
                break
    assert len(set(stopped.tolist())) > 1  # the batch really stops at different levels
    s = hip.BatchedLexLSE(batch, n, dims)
    s.setProblem(lod)
    s.setCtrType(types)
    s.factorize_solve()
    s.setSensitivityScan(True)
    found, ctr, obj, mx = s.ObjectiveSensitivity(start)
    np.testing.assert_array_equal(np.stack([found.astype(np.int32), ctr, obj], 1), sens)
    np.testing.assert_array_equal(mx, maxabs)
    np.testing.assert_array_equal(s.getWorkspace(), lam)
    np.testing.assert_array_equal(s.getCtrType(), marks)


def test_sensitivity_with_fixed_variables(hip, oracle):
    n, dims, batch = 12, [4, 4, 6], 6
    lod = P.lse_batch(51, batch, n, dims)
    nfixed = np.array([1, 2, 3, 0, 4, 2], np.uint32)
    idx = np.zeros((batch, n), np.uint32)
    val = np.zeros((batch, n))
    typ = np.zeros((batch, n), np.uint8)
    for b in range(batch):
        perm = np.argsort(P.uniform(800 + b, n))
        idx[b, :nfixed[b]] = perm[:nfixed[b]]
        val[b, :nfixed[b]] = P.normal(810 + b, n)[:nfixed[b]]
        typ[b, :nfixed[b]] = 1 + (P.uniform(820 + b, n)[:nfixed[b]] * 2).astype(np.uint8)
    types = np.full((batch, 14), 2, np.uint8)
    ref = oracle.lse_run(lod, dims, n, nfixed=nfixed, fixed_idx=idx, fixed_val=val, fixed_type=typ, ctr_type=types, sens_obj=2)
    s = hip.BatchedLexLSE(batch, n, dims)
    s.fixVariables(nfixed, idx, val, typ)
    s.setProblem(lod)
    s.setCtrType(types)
    s.factorize_solve()
    found, ctr, obj, maxabs = s.ObjectiveSensitivity(2)
    np.testing.assert_array_equal(np.stack([found.astype(np.int32), ctr, obj], 1), ref["sens"])
    np.testing.assert_array_equal(s.getWorkspace(), ref["lam"])


def test_round_blocks_match_separate_setters(hip, oracle):
    """lexls_lse_upload_round / download_round / sensitivity_resident (one copy each way per active-set round) give what the separate
    set_* / get_* calls give: dims, fixed variables, constraint types, skip flags and sensitivity levels from one host block."""
    n, dims, batch = 12, [4, 4, 6], 6
    lod = P.lse_batch(51, batch, n, dims)
    nfixed = np.array([1, 2, 3, 0, 4, 2], np.uint32)
    idx = np.zeros((batch, n), np.uint32)
    val = np.zeros((batch, n))
    typ = np.zeros((batch, n), np.uint8)
    for b in range(batch):
        perm = np.argsort(P.uniform(800 + b, n))
        idx[b, :nfixed[b]] = perm[:nfixed[b]]
        val[b, :nfixed[b]] = P.normal(810 + b, n)[:nfixed[b]]
        typ[b, :nfixed[b]] = 1 + (P.uniform(820 + b, n)[:nfixed[b]] * 2).astype(np.uint8)
    types = np.full((batch, 14), 2, np.uint8)
    ref = oracle.lse_run(lod, dims, n, nfixed=nfixed, fixed_idx=idx, fixed_val=val, fixed_type=typ, ctr_type=types, sens_obj=2)
    s = hip.BatchedLexLSE(batch, n, dims)
    L = s.round_layout()
    assert all(L[k] % 256 == 0 for k in L) and L["ctr_type"] > L["fixed_type"] > L["row_ld"]
    block = np.zeros(L["in_bytes"], np.uint8)
    v = s.round_views(block)
    v["dims"][:] = dims
    v["nfixed"][:], v["fixed_idx"][:], v["fixed_val"][:], v["fixed_type"][:] = nfixed, idx, val, typ
    v["ctr_type"][:] = types
    v["obj_index"][:] = 2
    v["skip"][3], v["obj_index"][3] = 1, -1  # problem 3 sits this round out
    s.setProblem(lod)
    s.upload_round(block)
    s.factorize_solve()
    s.sensitivity_resident()
    r = s.download_round()
    keep = np.arange(batch) != 3
    np.testing.assert_array_equal(r["x"][keep], ref["x"][keep])
    np.testing.assert_array_equal(r["total_rank"][keep], ref["totalrank"][keep])
    np.testing.assert_array_equal(r["found"][keep], ref["sens"][keep])
    np.testing.assert_array_equal(r["max_abs"][keep], ref["maxabs"][keep])
    np.testing.assert_array_equal(r["ctr_type"][keep], ref["ctr_type_out"][keep])
    np.testing.assert_array_equal(s.getWorkspace()[keep], ref["lam"][keep])
    assert (r["x"][3] == 0).all() and (r["ctr_type"][3] == 2).all()  # the skipped problem was not touched
    # argument checks of the separate setters are kept
    v["dims"][0, 0] = 99
    with pytest.raises(hip.LexlsError, match="exceeds the capacity"):
        s.upload_round(block)
    v["dims"][0, 0] = dims[0]
    v["fixed_idx"][0, 0] = n
    with pytest.raises(hip.LexlsError, match="out of range"):
        s.upload_round(block)


def test_least_norm_givens(hip, oracle):
    n, dims, batch = 40, [6] * 5, 8
    lod = P.lse_batch(61, batch, n, dims)
    ref = oracle.lse_run(lod, dims, n, solve_option=1)
    s = hip.BatchedLexLSE(batch, n, dims)
    s.setProblem(lod)
    s.factorize()
    s.solveLeastNorm_1()
    np.testing.assert_array_equal(s.get_x(), ref["x"])
    # least-norm property: x is orthogonal to the null space of the stacked constraints
    for b in range(batch):
        A = lod[b, :-1, :].T
        _, _, Vt = np.linalg.svd(A)
        assert np.abs(Vt[30:] @ s.get_x()[b]).max() < 1e-10


def test_least_norm_normal_equations(hip, oracle):
    """solveLeastNorm_2 (lexlse.h:1138-1213): bit-identical to the oracle; equal to the Givens variant within the tolerance the
    reference's MATLAB suites use (1e-10); with fixed variables and a rank-deficient level as well."""
    n, dims, batch = 40, [6] * 5, 8
    lod = P.lse_batch(61, batch, n, dims)
    ref = oracle.lse_run(lod, dims, n, solve_option=2)
    s = hip.BatchedLexLSE(batch, n, dims)
    s.setProblem(lod)
    s.factorize()
    s.solveLeastNorm_2()
    x2 = s.get_x().copy()
    np.testing.assert_array_equal(x2, ref["x"])
    s.solveLeastNorm_1()
    assert np.abs(s.get_x() - x2).max() < 1e-10
    # fixed variables + a duplicated row (rank deficiency)
    n, dims, batch = 12, [3, 4, 2], 5
    lod = P.lse_batch(62, batch, n, dims)
    lod[:, :, 4] = lod[:, :, 3]
    nfixed = np.full(batch, 2, np.uint32)
    idx = np.zeros((batch, n), np.uint32)
    idx[:, :2] = [5, 1]
    val = np.zeros((batch, n))
    val[:, :2] = P.normal(63, 2)
    ref = oracle.lse_run(lod, dims, n, solve_option=2, nfixed=nfixed, fixed_idx=idx, fixed_val=val)
    s = hip.BatchedLexLSE(batch, n, dims)
    s.fixVariables(nfixed, idx, val)
    s.setProblem(lod)
    s.factorize()
    s.solveLeastNorm_2()
    np.testing.assert_array_equal(s.get_x(), ref["x"])


@pytest.mark.parametrize("policy", [0, 1], ids=["wave-kernel", "generic-kernel"])
@pytest.mark.parametrize("reg_type", [1, 2, 3, 4, 5, 6, 7, 8, 9])
def test_regularization_family_bit_exact(hip, oracle, reg_type, policy):
    """lexlse.h:277-411 on the device — in the register-resident wave kernel's REG instantiation (default for these shapes: the routines of
    lexls_regularize.h work on the level's LDS image) and in the generic kernel: bit-identical to the oracle for every implemented type,
    on a hierarchy that takes both Tikhonov branches (tikhonov_1 and tikhonov_2), with fixed variables, per-problem factors and a
    rank-deficient level."""
    n, dims, batch = 12, [3, 4, 2], 6
    lod = P.lse_batch(7, batch, n, dims)
    lod[3:, :, 5] = lod[3:, :, 4]  # duplicated row in level 1 of half of the problems
    fac = np.abs(P.normal(88, batch * 3)).reshape(batch, 3) * 0.5 + 0.05
    fac[1, 1] = 0.0
    nfixed = np.array([0, 0, 2, 2, 1, 0], np.uint32)
    idx = np.zeros((batch, n), np.uint32)
    idx[:, :2] = [7, 2]
    val = np.zeros((batch, n))
    val[:, :2] = P.normal(89, 2)
    for b in range(batch):  # the oracle binding takes one factor vector per call
        ref = oracle.lse_run(lod[b:b + 1], dims, n, reg_type=reg_type, reg_factors=fac[b], nfixed=nfixed[b:b + 1], fixed_idx=idx[b:b + 1],
                             fixed_val=val[b:b + 1])
        s = hip.BatchedLexLSE(1, n, dims)
        s.set_kernel_policy(policy)
        s.setRegularization(reg_type, fac[b])
        s.fixVariables(nfixed[b:b + 1], idx[b:b + 1], val[b:b + 1])
        s.setProblem(lod[b:b + 1])
        s.factorize_solve()
        # (the experimental type 7 needs the level lists for its by-products: generic kernel whatever the policy)
        assert s.last_kernel().startswith("lqr_generic" if policy == 1 or reg_type == 7 else "lqr_wave<41,12,regularized>")
        assert_factor_equal(s, ref, dims, n)
        np.testing.assert_array_equal(s.get_x(), ref["x"])
        if reg_type == 7:
            xm, _, rm = s.get_mu()
            np.testing.assert_array_equal(xm, ref["x_mu"])
            np.testing.assert_array_equal(rm, ref["residual_mu"])
    # the whole batch at once with per-problem factors gives the same solutions
    s = hip.BatchedLexLSE(batch, n, dims)
    s.set_kernel_policy(policy)
    s.setRegularization(reg_type, fac)
    s.fixVariables(nfixed, idx, val)
    s.setProblem(lod)
    s.factorize_solve()
    for b in range(batch):
        ref = oracle.lse_run(lod[b:b + 1], dims, n, reg_type=reg_type, reg_factors=fac[b], nfixed=nfixed[b:b + 1], fixed_idx=idx[b:b + 1],
                             fixed_val=val[b:b + 1])
        np.testing.assert_array_equal(s.get_x()[b], ref["x"][0])


@pytest.mark.parametrize("reg_type", [1, 8, 3, 2])
def test_regularization_on_the_wave_kernel_ik_shapes(hip, oracle, reg_type):
    """the regularized wave kernel on the IK shape (n = 40, 5 x 12; columns run out in level 3: both Tikhonov branches) and on a wide one
    served by lqr_wave<64,16,regularized>, ragged levels included: factor, pivots, x bit-identical to the oracle, and to the generic kernel"""
    for (n, dims, seed, kernel) in ((40, [12] * 5, 4100, "lqr_wave<41,12,regularized>"), (55, [14, 9, 16, 5], 4200, "lqr_wave<64,16,regularized>")):
        batch = 5
        lod = P.lse_batch(seed, batch, n, dims)
        fac = np.abs(P.normal(seed + 1, batch * len(dims))).reshape(batch, len(dims)) * 0.3 + 0.01
        s = hip.BatchedLexLSE(batch, n, dims)
        s.setRegularization(reg_type, fac)
        s.setProblem(lod)
        s.factorize_solve()
        assert s.last_kernel() == kernel
        g = hip.BatchedLexLSE(batch, n, dims)
        g.set_kernel_policy(1)
        g.setRegularization(reg_type, fac)
        g.setProblem(lod)
        g.factorize_solve()
        np.testing.assert_array_equal(s.get_x(), g.get_x())
        np.testing.assert_array_equal(s.get_lexqr(), g.get_lexqr())
        for b in range(batch):
            ref = oracle.lse_run(lod[b:b + 1], dims, n, reg_type=reg_type, reg_factors=fac[b])
            np.testing.assert_array_equal(s.get_x()[b], ref["x"][0])
            np.testing.assert_array_equal(s.get_column_permutations()[b], ref["perm"][0])


@pytest.mark.parametrize("reg_type", [8, 1, 3])
def test_regularization_levels_the_reference_never_enters(hip, oracle, reg_type):
    """the reference returns before the first level when every variable is fixed (lexlse.h:164-175) and leaves the level loop once the columns are
    exhausted (:475-490): such levels are not regularized either — on the wave kernel as in the generic one (found by scripts/soak_lse.py:
    a problem with nVarFixed == nVar next to regular ones in one batch)"""
    n, cap_dims = 3, np.array([14, 16, 1, 13, 8, 1], np.uint32)
    dims = np.array([[3, 5, 1, 5, 5, 1], [14, 16, 1, 13, 8, 1], [2, 0, 1, 4, 8, 1]], np.uint32)
    batch = 3
    lod = np.zeros((batch, n + 1, int(cap_dims.sum())))
    for b in range(batch):
        m = int(dims[b].sum())
        lod[b, :, :m] = P.normal(7100 + b, (n + 1) * m).reshape(n + 1, m)
    nfixed = np.array([2, 3, 1], np.uint32)  # problem 1: all variables fixed
    idx = np.zeros((batch, n), np.uint32)
    idx[:, :3] = [2, 0, 1]
    val = np.zeros((batch, n))
    val[:, :3] = P.normal(7200, 3)
    fac = np.abs(P.normal(7300, 6)) * 0.3 + 0.01
    ref = oracle.lse_run(lod, dims, n, maxdim=cap_dims, nfixed=nfixed, fixed_idx=idx, fixed_val=val, reg_type=reg_type, reg_factors=fac)
    for policy in (0, 1):
        s = hip.BatchedLexLSE(batch, n, cap_dims)
        s.set_kernel_policy(policy)
        s.setObjDim(dims)
        s.setRegularization(reg_type, fac)
        s.fixVariables(nfixed, idx, val)
        s.setProblem(lod)
        s.factorize_solve(keep_factor=True)
        assert s.last_kernel().startswith("lqr_generic" if policy == 1 else "lqr_wave<64,16,regularized>")
        np.testing.assert_array_equal(s.get_x(), ref["x"])
        assert_factor_equal(s, ref, dims, n)


def test_regularization_variable_factor_and_least_norm_3(hip, oracle):
    n, dims, batch = 40, [6] * 5, 4
    lod = P.lse_batch(61, batch, n, dims)
    ref = oracle.lse_run(lod, dims, n, reg_type=1, reg_factors=[0.4] * 5, var_reg=1e3)
    s = hip.BatchedLexLSE(batch, n, dims)
    s.setRegularization(1, [0.4] * 5, variable_factor=1e3)
    s.setProblem(lod)
    s.factorize_solve()
    np.testing.assert_array_equal(s.get_x(), ref["x"])
    # solveLeastNorm_3: null-space basis of the Tikhonov family with zero factors (lexlse.h:1217-1221)
    ref3 = oracle.lse_run(lod, dims, n, solve_option=3, reg_type=1, reg_factors=[0.0] * 5)
    s.setRegularization(1, [0.0] * 5)
    s.factorize()
    s.solveLeastNorm_3()
    np.testing.assert_array_equal(s.get_x(), ref3["x"])
    s.solveLeastNorm_1()
    assert np.abs(s.get_x() - ref3["x"]).max() < 1e-10
    s.setRegularization(0)
    s.factorize()
    with pytest.raises(hip.LexlsError):
        s.solveLeastNorm_3()
    with pytest.raises(hip.LexlsError):
        s.setRegularization(10)  # not a LexLS::RegularizationType (typedefs.h:34-43)
    with pytest.raises(hip.LexlsError):
        s.get_mu()  # X_mu / residual_mu exist with REGULARIZATION_TIKHONOV_1 only


@pytest.mark.parametrize("n,dims", [(5, [2, 2]), (9, [3, 2, 3]), (23, [6, 5, 4]), (41, [7] * 4), (63, [9] * 5)])
def test_least_norm_3_with_an_odd_number_of_variables(hip, oracle, n, dims):
    """solveLeastNorm_3 ends in x = P x on an LDS scratch that must be 16-byte aligned whatever the parity of n (the copy of the
    solve kernel's permutation block computed the scratch from a base that is n doubles into the LDS block — misaligned for odd n)"""
    batch = 5
    lod = P.lse_batch(700 + n, batch, n, dims)
    ref3 = oracle.lse_run(lod, dims, n, solve_option=3, reg_type=1, reg_factors=[0.0] * len(dims))
    s = hip.BatchedLexLSE(batch, n, dims)
    s.setRegularization(1, [0.0] * len(dims))
    s.setProblem(lod)
    s.factorize()
    s.solveLeastNorm_3()
    np.testing.assert_array_equal(s.get_x(), ref3["x"])


def test_full_size_batch_4096(hip, oracle):
    """BASELINE.json configs[2]: batch 4096 x (n=40, 5x12) against the oracle on all problems — the batch bench.py times
    (problem id -> seed 20260100 + id)."""
    n, dims, batch = 40, [12] * 5, 4096
    lod = P.lse_batch(20260100, batch, n, dims)
    ref = oracle.lse_run(lod, dims, n, nthreads=8)
    s = hip.BatchedLexLSE(batch, n, dims)
    s.set_kernel_policy(4)  # the bit-exact four-per-wavefront kernel (automatic dispatch: the tolerance-contract one, tests/test_gpu_qtol.py)
    s.setProblem(lod)
    s.factorize_solve(keep_factor=False)
    assert s.last_kernel() == "lqr_quad<3,12,shift 7>"  # more problems than one round of the register-resident kernel holds: four per wavefront
    np.testing.assert_array_equal(s.get_x(), ref["x"])
    np.testing.assert_array_equal(s.get_column_permutations(), ref["perm"])
    small = hip.BatchedLexLSE(1024, n, dims)
    small.set_kernel_policy(4)
    small.setProblem(lod[:1024])
    small.factorize_solve(keep_factor=False)
    assert small.last_kernel() == "lqr_quad<3,12,shift 7>"  # x only: four per wavefront at every batch size (scripts/crossover.py)
    kept = hip.BatchedLexLSE(1024, n, dims)
    kept.setProblem(lod[:1024])
    kept.factorize_solve(keep_factor=True)
    assert kept.last_kernel() == "lqr_wave<41,12,exact>"  # factor kept, one round of it: the register-resident kernel
    np.testing.assert_array_equal(kept.get_x(), ref["x"][:1024])
    np.testing.assert_array_equal(small.get_x(), ref["x"][:1024])


def test_error_behaviour(hip):
    s = hip.BatchedLexLSE(2, 4, [2, 2])
    with pytest.raises(Exception):
        s.factorize()  # no problem data
    with pytest.raises(Exception):
        s.setObjDim([3, 2])  # exceeds capacity
    with pytest.raises(Exception):
        s.solve()  # no factorization


# --- the reference's manual lexlse suite on the device (interfaces/matlab-octave/tests/lexlse/test_lexlse_main.m) ---------------------
def device_lse_run(hip, lod, dims, n, reg_type=0, reg_factors=None, solve_option=0, nfixed=None, fixed_idx=None, fixed_val=None):
    """the call sequence of the MEX front end (interfaces/matlab-octave/lexlse.cpp:144-199) through the C ABI, one problem"""
    s = hip.BatchedLexLSE(lod.shape[0], n, dims)
    if reg_type:
        s.setRegularization(reg_type, np.asarray(reg_factors, float))
    if nfixed is not None:
        s.fixVariables(nfixed, fixed_idx, fixed_val)
    s.setProblem(lod)
    s.factorize()
    [s.solve, s.solveLeastNorm_1, s.solveLeastNorm_2, s.solveLeastNorm_3][solve_option]()
    return {"x": s.get_x()}


@pytest.mark.gpu
@pytest.mark.parametrize("least_norm,fixed_variables,reg_type,factors", P.lexlse_suite_options(),
                         ids=["ln%d-fix%d-type%d-f%d" % (o[0], o[1], o[2], o[3][0]) for o in P.lexlse_suite_options()])
def test_reference_lexlse_suite(hip, oracle, least_norm, fixed_variables, reg_type, factors):
    """The 42 option sets of test_lexlse_define.m (n = 30, m = [9,8,10,6], r = [7,6,8,5]) through the C ABI: the fixed-variable / least-norm
    formulation equals the general one with a terminal objective to the suite's 1e-10 (test_lexlse_main.m:20, compare_results.m), and both
    device solutions are bit-identical to the oracle's."""
    from test_oracle_golden import SUITE_N, SUITE_M, SUITE_R, SUITE_TOL, suite_solve
    n = SUITE_N
    for seed in (1, 2):
        blocks, fixed = P.lexlse_suite_problem(seed, n, SUITE_M, SUITE_R, bool(fixed_variables))
        gblocks, gfac = P.lexlse_suite_general_form(n, blocks, fixed, factors, least_norm)
        dev = lambda *a, **k: device_lse_run(hip, *a, **k)
        f1 = factors[1:] if fixed_variables else factors
        x1 = suite_solve(dev, n, blocks, fixed, reg_type, f1, least_norm)
        x2 = suite_solve(dev, n, gblocks, None, reg_type, gfac, 0)
        np.testing.assert_array_equal(x1, suite_solve(oracle.lse_run, n, blocks, fixed, reg_type, f1, least_norm))
        np.testing.assert_array_equal(x2, suite_solve(oracle.lse_run, n, gblocks, None, reg_type, gfac, 0))
        assert np.linalg.norm(x1 - x2) <= SUITE_TOL
        for g in gblocks:
            assert np.linalg.norm((g[:, :n] @ x1 - g[:, n]) - (g[:, :n] @ x2 - g[:, n])) <= SUITE_TOL


@pytest.mark.gpu
def test_tikhonov_1_sensitivity_and_byproducts(hip, oracle):
    """REGULARIZATION_TIKHONOV_1 (type 7, experimental in the reference): X_mu, residual_mu and — after ObjectiveSensitivity, which for
    this type starts from initialize_rhs and residual_mu (lexlse.h:647-651, :688-690, :1921-1959) — X_mu_rhs, the multipliers, the
    verdict and the CORRECT_SIGN marks are bit-identical to the oracle; rank-deficient and fixed-variable problems, a level with factor 0,
    a hierarchy whose columns run out (the copies of lexlse.h:483-486)."""
    cases = [(12, [4, 5, 3], None, [0.0, 0.7, 0.4], 0), (12, [4, 5, 6], None, [0.5, 0.7, 0.4], 2), (10, [3, 3, 3], [2, 3, 2], [0.3, 0.0, 0.6], 0),
             (9, [4, 5, 3, 2], None, [0.2, 0.3, 0.4, 0.5], 1)]
    for (n, dims, ranks, fac, nfix) in cases:
        batch = 3
        lod = np.stack([(P.rank_deficient_problem(30 + b, n, dims, ranks) if ranks else P.lse_problem(30 + b, n, dims)) for b in range(batch)])
        nfixed = np.full(batch, nfix, np.uint32)
        idx = np.zeros((batch, n), np.uint32)
        idx[:, :2] = [5, 1]
        val = np.zeros((batch, n))
        val[:, :2] = P.normal(31, 2)
        types = (np.arange(batch * sum(dims)).reshape(batch, -1) % 3 + 1).astype(np.uint8)  # LB / UB / EQ mix
        for k in range(len(dims)):
            ref = oracle.lse_run(lod, dims, n, reg_type=7, reg_factors=fac, sens_obj=k, nfixed=nfixed, fixed_idx=idx, fixed_val=val, ctr_type=types)
            s = hip.BatchedLexLSE(batch, n, dims)
            s.setRegularization(7, np.asarray(fac))
            s.fixVariables(nfixed, idx, val)
            s.setCtrType(types)
            s.setProblem(lod)
            s.factorize_solve()
            assert s.last_kernel().startswith("lqr_generic")
            np.testing.assert_array_equal(s.get_x(), ref["x"])
            found, ctr, obj, maxabs = s.ObjectiveSensitivity(k)
            xm, xr, rm = s.get_mu()
            np.testing.assert_array_equal(xm, ref["x_mu"])
            np.testing.assert_array_equal(rm, ref["residual_mu"])
            np.testing.assert_array_equal(xr, ref["x_mu_rhs"])
            np.testing.assert_array_equal(s.getWorkspace(), ref["lam"])
            np.testing.assert_array_equal(found.astype(np.int32), ref["sens"][:, 0])
            np.testing.assert_array_equal(ctr, ref["sens"][:, 1])
            np.testing.assert_array_equal(obj, ref["sens"][:, 2])
            np.testing.assert_array_equal(maxabs, ref["maxabs"])
            np.testing.assert_array_equal(s.getCtrType(), ref["ctr_type_out"])


@pytest.mark.gpu
@pytest.mark.parametrize("policy", [0, 1], ids=["default-kernel", "generic-kernel"])
def test_fixed_variable_index_chains_and_repeats(hip, oracle, policy):
    """fixVariable index bookkeeping of lexlse.h:132-153 (a later entry that names an already used slot is redirected to the column that
    slot's variable went to): long chains, a permutation of the leading slots, and the degenerate case of a variable named twice — the
    generic kernel resolves distinct indices through a look-up table and falls back to the reference's scan otherwise; everything
    bit-identical to the oracle."""
    n, dims = 70, [20, 30, 25]
    cases = [[5, 0, 1, 2, 3, 4], [3, 2, 1, 0], [1, 2, 3, 4, 5, 6, 7, 0], [9, 9, 4], [0, 5, 5, 0], list(range(20, 0, -1)), [2, 0, 2, 1]]
    batch = len(cases)
    lod = P.lse_batch(61, batch, n, dims)
    nfixed = np.array([len(c) for c in cases], np.uint32)
    idx, val = np.zeros((batch, n), np.uint32), np.zeros((batch, n))
    for b, c in enumerate(cases):
        idx[b, :len(c)] = c
        val[b, :len(c)] = P.normal(62 + b, len(c))
    ref = oracle.lse_run(lod, dims, n, nfixed=nfixed, fixed_idx=idx, fixed_val=val)
    s = hip.BatchedLexLSE(batch, n, dims)
    s.set_kernel_policy(policy)
    s.fixVariables(nfixed, idx, val)
    s.setProblem(lod)
    s.factorize_solve()
    assert s.last_kernel().startswith("lqr_generic")
    np.testing.assert_array_equal(s.get_column_permutations(), ref["perm"])
    np.testing.assert_array_equal(s.get_x(), ref["x"])
    assert_factor_equal(s, ref, dims, n)


@pytest.mark.gpu
@pytest.mark.parametrize("policy", [0, 5], ids=["step-per-pivot", "bit-exact-multi-launch"])
@pytest.mark.parametrize("empty", [3, 1, 0], ids=["last-level-empty", "middle-level-empty", "first-level-empty"])
def test_large_path_with_an_empty_level(hip, oracle, empty, policy):
    """A single problem on the large fast path whose level `empty` has no rows (capacity > 0, dimension 0): found by scripts/soak_lse.py
    (SOAK_WIDE=1) — the one-launch-per-level kernel was launched for the empty level and flipped the parity of the state buffers, which
    left x wrong while factor, pivots and ranks were right; an empty first or middle level made both large paths launch their Gauss-step
    kernel with zero threads (invalid configuration).  Pivots exact, values to 1e-10 on the fast path, bitwise on the other."""
    n, cap_dims = 188, np.array([149, 44, 131, 64], np.uint32)
    dims = np.array([[130, 40, 60, 30]], np.uint32)  # (more rows than the generic kernel's LDS image holds, whichever level is emptied)
    dims[0, empty] = 0
    m = int(dims.sum())
    lod = np.zeros((1, n + 1, int(cap_dims.sum())))
    lod[0, :, :m] = P.normal(4242 + empty, (n + 1) * m).reshape(n + 1, m)
    ref = oracle.lse_run(lod, dims, n, maxdim=cap_dims)
    s = hip.BatchedLexLSE(1, n, cap_dims)
    s.set_kernel_policy(policy)
    s.setObjDim(dims)
    s.setProblem(lod)
    s.factorize_solve()
    assert s.last_kernel().startswith("lqr_large<step-per-pivot" if policy == 0 else "lqr_large<multi-launch")
    np.testing.assert_array_equal(s.get_column_permutations(), ref["perm"])
    np.testing.assert_array_equal(s.getRanks()[0], ref["rank"])
    if policy == 5:
        np.testing.assert_array_equal(s.get_x(), ref["x"])
    assert np.abs(s.get_x() - ref["x"]).max() <= 1e-10


@pytest.mark.gpu
def test_large_path_row_that_repeats_an_earlier_level(hip, oracle):
    """The documented caveat of the step-per-pivot path (include/lexls_hip.h, kernel policy): a row that exactly repeats a row of an earlier
    level is rounding noise when its own level is reached; the sign of its reflector, beta = -sign(c0) |x|, follows the sign of that noise, and
    tree sums and ordered chains may disagree on it.  What the contract still guarantees, and what this test pins: pivots, ranks and first
    columns exact; x within 1e-10; every factor entry within 1e-10 (relative to the largest one) IN MAGNITUDE — a negated row of R with its
    negated essential part is the only admissible difference; the bit-exact policy 5 gives the oracle's factor, signs included."""
    n, dims = 150, [90, 90, 90]
    lod = P.lse_batch(77, 1, n, dims)
    lod[0, :, 90 + 7] = lod[0, :, 3]        # level 1, row 7 = level 0, row 3
    lod[0, :, 180 + 11] = lod[0, :, 90 + 20]  # level 2, row 11 = level 1, row 20
    ref = oracle.lse_run(lod, dims, n)
    cap = sum(dims)
    for policy in (0, 5):
        s = hip.BatchedLexLSE(1, n, dims)
        s.set_kernel_policy(policy)
        s.setProblem(lod)
        s.factorize_solve()
        assert s.last_kernel().startswith("lqr_large<step-per-pivot" if policy == 0 else "lqr_large<multi-launch")
        np.testing.assert_array_equal(s.get_column_permutations(), ref["perm"])
        r, fc, tr = s.getRanks()
        np.testing.assert_array_equal(r, ref["rank"])
        np.testing.assert_array_equal(fc, ref["fcol"])
        f, fr = s.get_lexqr()[0, :, :cap], ref["factor"][0, :, :cap]
        if policy == 5:
            np.testing.assert_array_equal(f, fr)
            np.testing.assert_array_equal(s.get_x(), ref["x"])
        else:
            scale = max(1.0, float(np.abs(fr).max()))
            assert np.abs(np.abs(f) - np.abs(fr)).max() <= 1e-10 * scale
            assert np.abs(s.get_x() - ref["x"]).max() <= 1e-10 * max(1.0, float(np.abs(ref["x"]).max()))
            # rows whose sign differs (if any) are whole rows of R: sign(f) = -sign(fr) on every entry of the row that is not noise
            differs = (np.sign(f) != np.sign(fr)) & (np.abs(fr) > 1e-9 * scale)
            rows = np.unique(np.nonzero(differs)[1])
            for row in rows:
                big = np.abs(fr[:, row]) > 1e-9 * scale
                pivot_and_right = big & (np.arange(n + 1) >= np.argmax(big))
                assert np.abs(f[pivot_and_right, row] + fr[pivot_and_right, row]).max() <= 1e-10 * scale, f"row {row}: not a negated row"


@pytest.mark.gpu
@pytest.mark.parametrize("keep", [False, True], ids=["x-only", "factor-kept"])
def test_deep_hierarchies_on_the_left_looking_kernels(hip, oracle, keep):
    """More than 64 rows in all (six to eight levels of an IK-sized problem) do not fit the register-resident kernel's LDS image of the rows
    below a level; the left-looking kernels read a level's rows when the level starts and serve them: four-per-wavefront kernel where its
    slots hold the columns, left-looking wave kernel otherwise — full rank, rank-deficient and ragged levels, fixed variables; bit-identical
    to the oracle and to the generic kernel."""
    cases = [(40, [12] * 8, None, 0, "lqr_quad<3,12,shift 7"), (40, [12] * 7, [12, 9, 6, 3, 3, 3, 2], 0, "lqr_quad<3,12,shift 7"), (30, [10, 12, 12, 11, 12, 9, 8], None, 3, "lqr_quad<3,12"),
             (47, [12] * 6, None, 0, "lqr_quad<3,12"), (55, [16, 14, 16, 12, 16], None, 0, "lqr_quad<4,16"), (63, [13, 16, 15, 16, 14], None, 2, "lqr_quad<4,16"), (36, [12] * 8, None, 2, "lqr_quad<3,12")]
    for (n, dims, ranks, nfix, kernel) in cases:
        batch = 9
        lod = np.stack([(P.rank_deficient_problem(5000 + b, n, dims, ranks) if ranks else P.lse_problem(5000 + b, n, dims)) for b in range(batch)])
        rdims = np.tile(np.array(dims, np.uint32), (batch, 1))
        rdims[1, 2] = 5  # a ragged problem inside the wavefront
        rdims[6, 0] = 0
        packed = np.zeros_like(lod)
        for b in range(batch):
            r = c = 0
            for k, d in enumerate(dims):
                packed[b, :, r:r + int(rdims[b, k])] = lod[b, :, c:c + int(rdims[b, k])]
                r += int(rdims[b, k])
                c += d
        kw = {}
        if nfix:
            idx, val = np.zeros((batch, n), np.uint32), np.zeros((batch, n))
            idx[:, :nfix] = [7, 2, 11][:nfix]
            val[:, :nfix] = P.normal(5100, nfix)
            kw = dict(nfixed=np.array([nfix] * (batch - 1) + [1], np.uint32), fixed_idx=idx, fixed_val=val)
        ref = oracle.lse_run(packed, rdims, n, maxdim=np.array(dims, np.uint32), **kw)
        s = hip.BatchedLexLSE(batch, n, dims)
        s.setObjDim(rdims)
        if nfix:
            s.fixVariables(kw["nfixed"], idx, val)
        s.setProblem(packed)
        s.factorize_solve(keep_factor=keep)
        assert s.last_kernel().startswith(kernel), (s.last_kernel(), kernel, n, dims)
        np.testing.assert_array_equal(s.get_x(), ref["x"])
        np.testing.assert_array_equal(s.get_column_permutations(), ref["perm"])
        np.testing.assert_array_equal(s.getRanks()[0], ref["rank"])
        if keep:
            assert_factor_equal(s, ref, rdims, n)
            np.testing.assert_array_equal(s.get_v(), ref["v"])


@pytest.mark.gpu
def test_wide_ik_shapes_keep_their_factor_on_the_quad_kernel_beyond_one_round(hip, oracle):
    """n + 1 in 49..64 or level dimensions 13..16 with the factor kept: up to one round of the register-resident kernel (2048 problems) that
    kernel, beyond it `lqr_quad<4,16,factor>` (1.4-1.6x faster there, scripts in DESIGN section 5) — same results, bit-identical to the oracle"""
    n, dims = 50, [14, 16, 13]
    for batch, kernel in ((2100, "lqr_quad<4,16,factor>"), (300, "lqr_wave<64,16>")):
        lod = P.lse_batch(6100, batch, n, dims)
        lod[7, :, 20] = lod[7, :, 3]  # a rank-deficient problem in the batch
        ref = oracle.lse_run(lod, dims, n, nthreads=8)
        s = hip.BatchedLexLSE(batch, n, dims)
        s.setProblem(lod)
        s.factorize_solve(keep_factor=True)
        assert s.last_kernel() == kernel
        np.testing.assert_array_equal(s.get_x(), ref["x"])
        np.testing.assert_array_equal(s.get_column_permutations(), ref["perm"])
        assert_factor_equal(s, ref, dims, n)
