"""Prefix reuse (the reference refactorizes everything in every LexLSI iteration, README.md:14, loop at lexlsi.h:1144-1172):
a factorization that reads the unchanged leading levels of its predecessor back instead of factorizing them
(lexls_lse_set_prefix_reuse / lexls_lse_set_resume_levels, lexls_amd/csrc/lqr_small_impl.h) must give EXACTLY what a full factorization gives —
factor, column permutation, ranks, first columns, Householder scalars and x, bit for bit — and what the CPU oracle gives."""
import numpy as np
import pytest

from lexls_amd import problems as P

pytestmark = pytest.mark.gpu

N, MAXD = 40, [12] * 5
CAP = sum(MAXD)


def pack(blocks_per_problem, n=N, cap=CAP):
    """[[rows of level k: (d_k, n + 1)]] per problem -> (lod (B, n + 1, cap), dims (B, nObj))"""
    B = len(blocks_per_problem)
    lod = np.zeros((B, n + 1, cap))
    dims = np.zeros((B, len(blocks_per_problem[0])), np.uint32)
    for b, blocks in enumerate(blocks_per_problem):
        rows = np.vstack([blk for blk in blocks if len(blk)]) if any(len(blk) for blk in blocks) else np.zeros((0, n + 1))
        lod[b, :, :rows.shape[0]] = rows.T
        dims[b] = [len(blk) for blk in blocks]
    return lod, dims


def random_blocks(seed, dims, n=N):
    out, st = [], 0
    for d in dims:
        out.append(P.normal(seed, d * (n + 1), st).reshape(d, n + 1))
        st += 1
    return out


def change_level(blocks, K, seed, how, n=N, maxd=12):
    """the LexLSI moves on level K: a row appended (activation, workingset.h:79-92), a row erased in order (deactivation, :99-108), or
    one row's data replaced (a bound that switches sides); the levels below may change as well"""
    out = [blk.copy() for blk in blocks]
    if K >= len(out):
        return out
    blk = out[K]
    if how == 0 and len(blk) < maxd:
        out[K] = np.vstack([blk, P.normal(seed, n + 1, 90)[None]])
    elif how == 1 and len(blk) > 1:
        out[K] = np.delete(blk, (seed % len(blk)), axis=0)
    else:
        if len(blk):
            out[K][seed % len(blk)] = P.normal(seed, n + 1, 91)
    if seed % 3 == 0 and K + 1 < len(out) and len(out[K + 1]):  # something else further down
        out[K + 1][0] = P.normal(seed, n + 1, 92)
    return out


def factor_all(s):
    r, fc, tr = s.getRanks()
    return dict(x=s.get_x(), factor=s.get_lexqr(), hh=s.get_hh_scalars(), perm=s.get_column_permutations(), rank=r, fcol=fc, totalrank=tr)


def run_pair(hip, oracle, blocks1, blocks2, K, fixed=None, n=N, maxd=MAXD, policy=2):
    """factorize P1, then P2 with the levels K[b] read back; a fresh handle factorizes P2 from scratch"""
    cap = sum(maxd)
    lod1, dims1 = pack(blocks1, n, cap)
    lod2, dims2 = pack(blocks2, n, cap)
    B = lod1.shape[0]

    def make():
        s = hip.BatchedLexLSE(B, n, maxd)
        s.set_kernel_policy(policy)
        if fixed is not None:
            s.fixVariables(*fixed)
        return s

    s = make()
    s.set_prefix_reuse(True)
    s.setObjDim(dims1)
    s.setProblem(lod1)
    s.factorize_solve(keep_factor=True)
    assert s.last_kernel().startswith("lqr_wave<") and s.prefix_reuse_ready()
    s.setObjDim(dims2)
    s.setProblem(lod2)
    s.set_resume_levels(K)
    s.factorize_solve(keep_factor=True)
    got = factor_all(s)
    assert s.prefix_reuse_ready()

    r = make()
    r.setObjDim(dims2)
    r.setProblem(lod2)
    r.factorize_solve(keep_factor=True)
    ref = factor_all(r)
    for key in ref:
        if key in ("factor", "hh"):  # (rows beyond a problem's own are not written)
            for b in range(B):
                m = int(dims2[b].sum())
                np.testing.assert_array_equal(got[key][b][..., :m], ref[key][b][..., :m], err_msg=f"{key}, problem {b} (levels read back: {K[b]})")
        else:
            np.testing.assert_array_equal(got[key], ref[key], err_msg=key)
    if fixed is None:
        o = oracle.lse_run(lod2, dims2, n, maxdim=np.asarray(maxd, np.uint32))
        np.testing.assert_array_equal(got["x"], o["x"])
        np.testing.assert_array_equal(got["perm"], o["perm"])
        np.testing.assert_array_equal(got["rank"], o["rank"])
    return s, got


@pytest.mark.parametrize("seed", [1, 2, 3])
def test_one_level_changes_per_problem(hip, oracle, seed):
    """ragged IK-sized problems; every problem changes at its own level (0 .. nObj: 0 = everything again, nObj = nothing changed)"""
    B = 97
    rng = np.random.default_rng(seed)
    b1, b2, K = [], [], []
    for b in range(B):
        dims = rng.integers(5, 13, size=5)
        blocks = random_blocks(1000 * seed + b, dims)
        k = int(rng.integers(0, 6))
        b1.append(blocks)
        b2.append(change_level(blocks, k, 7000 * seed + b, int(rng.integers(0, 3))))
        K.append(k)
    run_pair(hip, oracle, b1, b2, np.array(K, np.int32))


def test_rank_deficient_levels_and_exhausted_columns(hip, oracle):
    """levels with exact linear dependence (rank break, lexlse.h:214) above and below the changed one; few variables: the columns run out
    inside a level that is read back"""
    B = 40
    b1, b2, K = [], [], []
    for b in range(B):
        lod = P.rank_deficient_problem(300 + b, N, MAXD, [7, 12, 5, 9, 12])
        blocks = [np.hstack([A, rhs[:, None]]) for A, rhs in P.levels_of(lod, MAXD)]
        k = b % 6
        b1.append(blocks)
        b2.append(change_level(blocks, k, 900 + b, b % 3))
        K.append(k)
    run_pair(hip, oracle, b1, b2, np.array(K, np.int32))
    n = 20
    b1, b2, K = [], [], []
    for b in range(B):
        blocks = random_blocks(5000 + b, [12, 12, 12], n)
        k = b % 4
        b1.append(blocks)
        b2.append(change_level(blocks, k, 5100 + b, b % 3, n))
        K.append(k)
    run_pair(hip, oracle, b1, b2, np.array(K, np.int32), n=n, maxd=[12] * 3)


def test_with_fixed_variables(hip, oracle):
    """fixed variables (the simple-bounds objective of a LexLSI problem, lexlse.h:132-156) stay what they were; general levels change"""
    B = 33
    rng = np.random.default_rng(5)
    nfixed = rng.integers(0, 4, size=B).astype(np.uint32)
    index = np.zeros((B, N), np.uint32)
    value = np.zeros((B, N))
    for b in range(B):
        index[b, :nfixed[b]] = rng.choice(N, size=nfixed[b], replace=False)
        value[b, :nfixed[b]] = rng.normal(size=nfixed[b])
    b1, b2, K = [], [], []
    for b in range(B):
        blocks = random_blocks(8000 + b, rng.integers(4, 13, size=5))
        k = int(rng.integers(0, 6))
        b1.append(blocks)
        b2.append(change_level(blocks, k, 8100 + b, b % 3))
        K.append(k)
    run_pair(hip, oracle, b1, b2, np.array(K, np.int32), fixed=(nfixed, index, value))


def test_chain_of_changes(hip, oracle):
    """a LexLSI-like sequence on ONE handle: ten factorizations, each resuming from the one before at a fresh level"""
    B = 16
    rng = np.random.default_rng(11)
    cur = [random_blocks(100 + b, rng.integers(6, 12, size=5)) for b in range(B)]
    s = hip.BatchedLexLSE(B, N, MAXD)
    s.set_kernel_policy(2)
    s.set_prefix_reuse(True)
    lod, dims = pack(cur)
    s.setObjDim(dims)
    s.setProblem(lod)
    s.factorize_solve(keep_factor=True)
    for it in range(10):
        K = rng.integers(0, 6, size=B).astype(np.int32)
        cur = [change_level(cur[b], int(K[b]), 40 * it + b, int(rng.integers(0, 3))) for b in range(B)]
        lod, dims = pack(cur)
        s.setObjDim(dims)
        s.setProblem(lod)
        s.set_resume_levels(K)
        s.factorize_solve(keep_factor=True)
        o = oracle.lse_run(lod, dims, N, maxdim=np.asarray(MAXD, np.uint32))
        np.testing.assert_array_equal(s.get_x(), o["x"])
        np.testing.assert_array_equal(s.get_column_permutations(), o["perm"])
        np.testing.assert_array_equal(s.getRanks()[0], o["rank"])
        f = s.get_lexqr()
        for b in range(B):
            m = int(dims[b].sum())
            np.testing.assert_array_equal(f[b, :, :m], o["factor"][b, :, :m])


def test_requests_that_cannot_be_served(hip):
    """nothing to resume from after an x-only solve or another kernel: the request is refused, never silently wrong"""
    lod = P.lse_batch(3, 8, N, MAXD)
    s = hip.BatchedLexLSE(8, N, MAXD)
    s.setProblem(lod)
    with pytest.raises(hip.LexlsError):
        s.set_resume_levels(np.zeros(8, np.int32))  # not enabled
    s.set_prefix_reuse(True)
    with pytest.raises(hip.LexlsError):
        s.set_resume_levels(np.zeros(8, np.int32))  # no factorization yet
    s.factorize_solve(keep_factor=False)
    assert not s.prefix_reuse_ready()
    s.set_kernel_policy(1)
    s.factorize_solve(keep_factor=True)
    assert not s.prefix_reuse_ready()  # the generic kernel leaves no state
    s.set_kernel_policy(2)
    s.factorize_solve(keep_factor=True)
    assert s.prefix_reuse_ready()
    with pytest.raises(hip.LexlsError):
        s.set_resume_levels(np.full(8, 9, np.int32))  # more levels than the problem has
