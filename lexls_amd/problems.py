"""Deterministic synthetic lexicographic least-squares problems.

Counter-based generator: splitmix64 finaliser -> uniform(0,1) -> Box-Muller normal, so every host
(this container, the GPU box) produces bit-identical inputs from a seed; no numpy RNG state.

Layouts follow the reference's equality solver storage (lexlse.h:85): one problem = a column-major
``cap x (nVar+1)`` array ``LOD`` whose rows are the stacked levels ``[A_k | b_k]`` (column nVar =
right-hand side); a batch is ``(batch, nVar+1, cap)`` in C order, i.e. each problem column-major.
"""
from __future__ import annotations

import numpy as np

_M64 = np.uint64(0xFFFFFFFFFFFFFFFF)


def _mix(z: np.ndarray) -> np.ndarray:
    """splitmix64 output function on an array of uint64 counters."""
    with np.errstate(over="ignore"):
        z = (z + np.uint64(0x9E3779B97F4A7C15)) & _M64
        z = ((z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)) & _M64
        z = ((z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)) & _M64
        return z ^ (z >> np.uint64(31))


def uniform(seed: int, count: int, stream: int = 0) -> np.ndarray:
    """`count` doubles in (0,1), a pure function of (seed, stream, index)."""
    with np.errstate(over="ignore"):
        base = _mix(np.array([seed], dtype=np.uint64) * np.uint64(0x632BE59BD9B4E019) + np.uint64(stream))[0]
        ctr = np.arange(count, dtype=np.uint64) * np.uint64(0x9E3779B97F4A7C15) + base
    bits = _mix(ctr) >> np.uint64(11)  # 53 random bits
    return (bits.astype(np.float64) + 0.5) * (1.0 / 9007199254740992.0)


def normal(seed: int, count: int, stream: int = 0) -> np.ndarray:
    """`count` N(0,1) doubles (Box-Muller on pairs of uniforms)."""
    m = (count + 1) // 2
    u1 = uniform(seed, m, 2 * stream)
    u2 = uniform(seed, m, 2 * stream + 1)
    r = np.sqrt(-2.0 * np.log(u1))
    out = np.empty(2 * m)
    out[0::2] = r * np.cos(2.0 * np.pi * u2)
    out[1::2] = r * np.sin(2.0 * np.pi * u2)
    return out[:count]


def lse_problem(seed: int, nvar: int, dims, cap_dims=None) -> np.ndarray:
    """One iid-N(0,1) equality problem; returns LOD as an (nVar+1, cap) C array (= column-major cap x (nVar+1))."""
    dims = list(dims)
    cap_dims = list(cap_dims) if cap_dims is not None else dims
    cap, m = sum(cap_dims), sum(dims)
    vals = normal(seed, m * (nvar + 1)).reshape(nvar + 1, m)
    lod = np.zeros((nvar + 1, cap))
    lod[:, :m] = vals
    return lod


def lse_batch(seed0: int, batch: int, nvar: int, dims, cap_dims=None) -> np.ndarray:
    """Batch of iid problems, problem b uses seed0 + b (C3 seed rule). Shape (batch, nVar+1, cap)."""
    return np.stack([lse_problem(seed0 + b, nvar, dims, cap_dims) for b in range(batch)])


def lse_batch_fast(seed0: int, batch: int, nvar: int, dims) -> np.ndarray:
    """Same distribution as lse_batch but generated in one vectorised call (bench-sized batches)."""
    m = sum(dims)
    per = m * (nvar + 1)
    vals = normal(seed0, batch * per)
    return vals.reshape(batch, nvar + 1, m).copy()


def rank_deficient_problem(seed: int, nvar: int, dims, ranks) -> np.ndarray:
    """Problem whose level k has exactly `ranks[k]` new directions beyond the levels above it.

    Construction of the reference's random-test generator
    (interfaces/matlab-octave/tests/implementation/utility/define_problem.m:29-55):
    A_k = randn(m_k, sum(m_<k) + r_k) @ [C; randn(r_k, n)], C = previous levels rescaled by max|C|.
    Exact linear dependence exercises the rank `break` (lexlse.h:214).
    """
    dims, ranks = list(dims), list(ranks)
    C = np.zeros((0, nvar))
    blocks = []
    stream = 0
    for m_k, r_k in zip(dims, ranks):
        g = normal(seed, m_k * (C.shape[0] + r_k), stream).reshape(m_k, C.shape[0] + r_k)
        fresh = normal(seed, r_k * nvar, stream + 1).reshape(r_k, nvar)
        b = normal(seed, m_k, stream + 2)
        stream += 3
        A = g @ np.vstack([C, fresh])
        blocks.append(np.hstack([A, b[:, None]]))
        C = np.vstack([C, A])
        s = np.abs(C).max() if C.size else 0.0
        if s > 1:
            C = C / s
    stacked = np.vstack(blocks)  # (M, nVar+1)
    return np.ascontiguousarray(stacked.T)


def lexlse_suite_problem(seed: int, nvar: int, dims, ranks, fixed_variables: bool):
    """The random problem of the reference's manual lexlse suite (interfaces/matlab-octave/tests/lexlse/define_problem.m:29-55):
    as rank_deficient_problem, but with `fixed_variables` the first level fixes dims[0] distinct variables to random values.
    Returns (blocks, fixed): blocks = [A_k | b_k] of the general levels, fixed = (indices, values) or None."""
    dims, ranks = list(dims), list(ranks)
    C = np.zeros((0, nvar))
    blocks, fixed, stream = [], None, 0
    for k, (m_k, r_k) in enumerate(zip(dims, ranks)):
        if fixed_variables and k == 0:
            idx = np.argsort(uniform(seed, nvar, 1000), kind="stable")[:m_k]
            A = np.zeros((m_k, nvar))
            A[np.arange(m_k), idx] = 1.0
            fixed = (idx.astype(np.uint32), normal(seed, m_k, 1001))
            C = np.vstack([C, A])
            continue
        g = normal(seed, m_k * (C.shape[0] + r_k), stream).reshape(m_k, C.shape[0] + r_k)
        fresh = normal(seed, r_k * nvar, stream + 1).reshape(r_k, nvar)
        b = normal(seed, m_k, stream + 2)
        stream += 3
        A = g @ np.vstack([C, fresh])
        blocks.append(np.hstack([A, b[:, None]]))
        C = np.vstack([C, A])
        s = np.abs(C).max()
        if s > 1:
            C = C / s
    return blocks, fixed


def lexlse_suite_options():
    """The 42 option sets of test_lexlse_define.m: (get_least_norm_solution, enable_fixed_variables, regularization_type, factors).
    Types: TIKHONOV 1, TIKHONOV_1 7, TIKHONOV_2 8 with least-norm options 0-3; R 3, R_NO_Z 4, RT_NO_Z 5 with options 0-2."""
    out = []
    for t in (1, 7, 8):
        out.append((0, 0, t, (1, 2, 3, 4)))
        out += [(ln, fx, t, (0, 2, 3, 4)) for ln, fx in ((0, 1), (1, 1), (1, 0), (2, 1), (2, 0), (3, 1), (3, 0))]
    for t in (3, 4, 5):
        out += [(ln, fx, t, (0, 2, 3, 4)) for ln in (0, 1, 2) for fx in (1, 0)]
    return out


def lexlse_suite_general_form(nvar: int, blocks, fixed, factors, least_norm: int):
    """fixed2general.m + append_terminal_objective.m: the fixed variables as a first level of unit rows, and — when a least-norm solution
    is asked for — a terminal level I x = 0 with factor 0.  Returns (blocks, factors) of the equivalent general problem, which is solved
    with solve()."""
    gblocks, gfac = list(blocks), list(factors)
    if fixed is not None:
        A1 = np.zeros((len(fixed[0]), nvar + 1))
        A1[np.arange(len(fixed[0])), fixed[0]] = 1.0
        A1[:, nvar] = fixed[1]
        gblocks = [A1] + gblocks
    if least_norm:
        gblocks = gblocks + [np.hstack([np.eye(nvar), np.zeros((nvar, 1))])]
        gfac = gfac + [0.0]
    return gblocks, gfac


def stack_levels(blocks) -> np.ndarray:
    """[A_k | b_k] blocks -> one (nVar+1, cap) problem in the layout of this package."""
    return np.ascontiguousarray(np.vstack(blocks).T)


def levels_of(lod: np.ndarray, dims):
    """Split an (nVar+1, cap) problem into [(A_k, b_k)]."""
    out, r = [], 0
    for d in dims:
        blk = lod[:, r:r + d].T
        out.append((blk[:, :-1].copy(), blk[:, -1].copy()))
        r += d
    return out


# --- flop / byte model of one factorize()+solve() -------------------------

def flop_model(nvar: int, dims, ranks=None) -> dict:
    """Algorithmic flops of one l-QR + back-substitution, FMA = 2 flops; `ranks` defaults to generic full rank."""
    n = nvar
    dims = list(dims)
    if ranks is None:
        ranks, left = [], n
        for d in dims:
            r = min(d, left)
            ranks.append(r)
            left -= r
    M = sum(dims)
    out = dict(norm0=0, pivot=0, make=0, apply=0, downdate=0, trsm=0, gemm=0, solve=0)
    c, F, acc_tail = 0, 0, []
    for k, (m, r) in enumerate(zip(dims, ranks)):
        if c < n:
            out["norm0"] += 2 * m * (n - c)
        for j in range(r):
            R, C = m - j, n - c - j
            out["pivot"] += C + 2 * R
            if R > 1:
                out["make"] += 3 * (R - 1) + 6
                out["apply"] += 4 * (R - 1) * C + 4 * C
            out["downdate"] += 2 * max(C - 1, 0)
        below = M - (F + m)
        if k < len(dims) - 1 and r > 0:
            out["trsm"] += below * r * r
            out["gemm"] += 2 * below * r * (n - (c + r) + 1)
        c += r
        F += m
    acc = 0
    for r in reversed(ranks):
        if r > 0:
            out["solve"] += 2 * r * acc + r * r
            acc += r
    out["total"] = sum(out.values())
    return out


def algorithmic_bytes(nvar: int, dims, write_factor: bool = False) -> int:
    """Bytes one factorize()+solve() must move: read [A|b] once, write x (+ the factor variant), SURVEY 8(d)."""
    M = sum(dims)
    b = 8 * M * (nvar + 1) + 8 * nvar
    if write_factor:
        b += 8 * M * (nvar + 1) + 8 * M + 4 * nvar + 8 * len(list(dims))
    return b


def _levels_reached_x_only(nvar: int, dims, ranks=None):
    """Number of leading levels an x-only solve has to read: those that start before the columns are exhausted."""
    dims = list(dims)
    if ranks is None:
        ranks, left = [], nvar
        for d in dims:
            r = min(d, left)
            ranks.append(r)
            left -= r
    c, reached = 0, 0
    for r in ranks:
        if c >= nvar:
            break
        reached += 1
        c += r
    return reached, ranks


def bytes_touched_x_only(nvar: int, dims, ranks=None) -> int:
    """Bytes the x-only kernels actually move per problem: the rows of the levels reached before the columns are exhausted (the rows below
    never influence x), x, the column permutation, ranks / first columns, TotalRank."""
    reached, _ = _levels_reached_x_only(nvar, dims, ranks)
    dims = list(dims)
    return 8 * sum(dims[:reached]) * (nvar + 1) + 8 * nvar + 4 * nvar + 8 * len(dims) + 4


def flops_executed_x_only(nvar: int, dims, ranks=None) -> int:
    """flop_model() without the elimination of the rows of the levels that are never reached (their multipliers and Schur complements are
    part of the factor, not of x)."""
    reached, ranks = _levels_reached_x_only(nvar, dims, ranks)
    dims = list(dims)
    total = flop_model(nvar, dims, ranks)["total"]
    skipped_rows = sum(dims[reached:])
    c = 0
    for k, r in enumerate(ranks[:reached]):
        if k < len(dims) - 1 and r > 0:
            total -= skipped_rows * r * r + 2 * skipped_rows * r * (nvar - (c + r) + 1)
        c += r
    return total


# --- inequality (LexLSI) problems: BASELINE.md C5 -------------------------------------------------

def lsi_problem(seed: int, nvar: int = 40, dims=(12, 12, 12, 12, 12), simple_bounds: bool = True, perturb: float = 0.0, perturb_seed: int = 0):
    """Objectives of one LexLSI problem: level 0 = simple bounds -1 <= x_i <= 1 on dims[0] variables, middle levels
    b - w <= A x <= b + w with w ~ U(0,1), last level equalities A x = b.  `perturb` adds perturb * N(0,1) to every b
    (the warm-start neighbour of BASELINE.md C5)."""
    dims = list(dims)
    objs, stream = [], 0
    for k, m in enumerate(dims):
        if k == 0 and simple_bounds:
            var = np.argsort(uniform(seed, nvar, 1000))[:m].astype(np.uint32)
            objs.append(dict(var=var, lb=-np.ones(m), ub=np.ones(m)))
            continue
        A = normal(seed, m * nvar, stream).reshape(m, nvar)
        b = normal(seed, m, stream + 1)
        w = uniform(seed, m, 2000 + k)
        stream += 2
        if perturb:
            b = b + perturb * normal(seed + 7919 * (perturb_seed + 1), m, 3000 + k)
        if k == len(dims) - 1:
            objs.append(dict(A=A, lb=b.copy(), ub=b.copy()))
        else:
            objs.append(dict(A=A, lb=b - w, ub=b + w))
    return objs
