"""Development aid for lqr_mfma (lexls_amd/csrc/lqr_mfma_impl.h): small batches against the oracle with a per-problem report, then timings.
Run on the GPU box: python3 scripts/dbg_mfma.py [policies...]"""
import sys, time, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import lexls_amd as hip
from lexls_amd import problems as P
from oracle import oracle_ctypes as oracle

N, DIMS = 40, [12] * 5


def report(tag, lod, dims, n, policy):
    ref = oracle.lse_run(lod, dims, n, nthreads=8)
    s = hip.BatchedLexLSE(lod.shape[0], n, dims)
    s.set_kernel_policy(policy)
    s.setProblem(lod)
    s.factorize_solve(keep_factor=False)
    r, fc, tr = s.getRanks()
    perm = s.get_column_permutations()
    x = s.get_x()
    okr = (r == ref["rank"]).all(axis=1)
    okp = (perm == ref["perm"]).all(axis=1)
    err = np.abs(x - ref["x"]).max(axis=1) / max(1.0, float(np.abs(ref["x"]).max()))
    bad = np.where(~okr | ~okp | ~(err <= 1e-10))[0]
    print(f"[{tag}] policy {policy} kernel {s.last_kernel()} batch {lod.shape[0]}: ranks ok {okr.sum()}, perm ok {okp.sum()}, max x err {np.nanmax(err):.3e}, nan {np.isnan(x).any()}, bad {len(bad)}", flush=True)
    for b in bad[:3]:
        print("  problem", b, "rank", r[b], "ref", ref["rank"][b])
        d = np.where(perm[b] != ref["perm"][b])[0]
        print("   first perm mismatch at", d[:5], "got", perm[b][d[:5]], "ref", ref["perm"][b][d[:5]])
        print("   perm got", perm[b][:16], "\n   perm ref", ref["perm"][b][:16])
        print("   x got", x[b][:6], "\n   x ref", ref["x"][b][:6], " err", err[b])
    return len(bad) == 0


def timing(policy, batch=4096, reps=200):
    import ctypes as C
    lods = [P.lse_batch_fast(20260100 + i, batch, N, DIMS) for i in range(2)]
    s = hip.BatchedLexLSE(batch, N, DIMS)
    s.set_kernel_policy(policy)
    s.setProblem(lods[0])
    for _ in range(20):
        s.factorize_solve(keep_factor=False)
    t0 = time.perf_counter()
    for _ in range(reps):
        s.factorize_solve(keep_factor=False)
    s.synchronize() if hasattr(s, "synchronize") else None
    s.get_x()
    dt = (time.perf_counter() - t0) / reps
    print(f"policy {policy} kernel {s.last_kernel()}: {dt*1e6:.1f} us per batch of {batch} (host-timed, includes launch gaps)", flush=True)


if __name__ == "__main__":
    pols = [int(v) for v in sys.argv[1:]] or [7, 8]
    for pol in pols:
        ok = True
        ok &= report("ik-1", P.lse_batch(1, 1, N, DIMS), DIMS, N, pol)
        ok &= report("ik-5", P.lse_batch(2, 5, N, DIMS), DIMS, N, pol)
        ok &= report("ik-64", P.lse_batch(3, 64, N, DIMS), DIMS, N, pol)
        ok &= report("one-level", P.lse_batch(4, 8, N, [12]), [12], N, pol)
        ok &= report("two-level", P.lse_batch(5, 8, N, [12, 12]), [12, 12], N, pol)
        ok &= report("n=20", P.lse_batch(6, 9, 20, [12] * 3), [12] * 3, 20, pol)
        ok &= report("n=47", P.lse_batch(7, 9, 47, [12] * 5), [12] * 5, 47, pol)
        rd = np.stack([P.rank_deficient_problem(300 + b, N, DIMS, [9, 12, 7, 12, 12]) for b in range(21)])
        ok &= report("rank-def", rd, DIMS, N, pol)
        ok &= report("ik-4096", P.lse_batch_fast(20260100, 4096, N, DIMS), DIMS, N, pol)
        print("policy", pol, "ALL OK" if ok else "FAILURES", flush=True)
    for pol in [6] + pols:
        timing(pol)
