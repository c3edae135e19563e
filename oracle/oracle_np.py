"""ORACLE — TEST INFRASTRUCTURE ONLY.  Independent numpy cross-check that does NOT follow the l-QR
algorithm: sequential null-space projection (the reference's own cross-check idea,
interfaces/matlab-octave/tests/lexlse/seq_lexls.m and examples/example_lexlse.m:17-29).

For levels (A_k, b_k), k = 1..P:   x <- x + N z,  z = pinv(A_k N)(b_k - A_k x),  N <- N null(A_k N).
The per-level optimal residual norms ||A_k x - b_k|| are unique; x itself is unique only when the
stacked levels have full column rank (otherwise this returns the step-wise least-norm solution).
"""
import numpy as np


def lex_solve(levels, rtol=1e-10):
    n = levels[0][0].shape[1]
    x = np.zeros(n)
    N = np.eye(n)
    ranks = []
    for A, b in levels:
        if N.shape[1] == 0:
            ranks.append(0)
            continue
        AN = A @ N
        U, s, Vt = np.linalg.svd(AN, full_matrices=True)
        thr = rtol * max(s[0], 1.0) if s.size else 0.0
        r = int((s > thr).sum())
        ranks.append(r)
        if r > 0:
            z = Vt[:r].T @ ((U[:, :r].T @ (b - A @ x)) / s[:r])
            x = x + N @ z
        N = N @ Vt[r:].T
    res = [float(np.linalg.norm(A @ x - b)) for A, b in levels]
    return x, res, ranks


def residual_norms(levels, x):
    return [float(np.linalg.norm(A @ x - b)) for A, b in levels]
