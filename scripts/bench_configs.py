#!/usr/bin/env python3
"""Secondary measurements (not the headline bench.py line): the other BASELINE configs and the secondary kernels.
Prints one JSON object; run on the GPU box:  python scripts/bench_configs.py > gpurun_out/configs.json"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import lexls_amd  # noqa: E402
from lexls_amd import lexlsi, problems as P  # noqa: E402
from oracle import oracle_ctypes as oc  # noqa: E402


def timed(fn, sync, reps):
    fn(); sync()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    sync()
    return (time.perf_counter() - t0) / reps


out = {}

# ---- configs[1]: single large equality problem n=512, 4 x 256 -------------------------------------------------------
n, dims = 512, [256] * 4
lod = P.lse_batch(20260001, 1, n, dims)
s = lexls_amd.BatchedLexLSE(1, n, dims)
s.setProblem(lod)
t = timed(lambda: s.factorize_solve(True), s.synchronize, 5)
tf = timed(lambda: s.factorize(), s.synchronize, 5)
flops = P.flop_model(n, dims)["total"]
tc, _ = oc.lse_time(lod, dims, n, 1, 3)
s5 = lexls_amd.BatchedLexLSE(1, n, dims)
s5.set_kernel_policy(5)
s5.setProblem(lod)
t5 = timed(lambda: s5.factorize_solve(True), s5.synchronize, 5)
out["config1_single_large"] = dict(kernel=s.last_kernel(), ms=1e3 * t, factorize_only_ms=1e3 * tf, gflops=flops / t / 1e9, cpu_oracle_ms=1e3 * tc / 3, cpu_oracle_gflops=flops / (tc / 3) / 1e9,
                                   bit_exact_path=dict(kernel=s5.last_kernel(), ms=1e3 * t5),
                                   note="fast path: the pivots of a level in one launch (tagged-granule hand-offs between 129 workgroups), tree sums, trailing update on v_mfma_f64_16x16x4; "
                                        "bit_exact_path = ordered chains, two launches per pivot (policy 5)")

# ---- secondary kernels on the IK batch ---------------------------------------------------------------------------------
n, dims, batch = 40, [12] * 5, 4096
lod = P.lse_batch_fast(20260100, batch, n, dims)
s = lexls_amd.BatchedLexLSE(batch, n, dims)
s.setProblem(lod)
s.setCtrType(np.full((batch, 60), 2, np.uint8))
s.factorize_solve(True)
fac_bytes = batch * 60 * 41 * 8
import ctypes as C
from lexls_amd import capi
L = capi.lib()
for name, fn in [("residual(get_v)", lambda: L.lexls_lse_residual(s._h)),
                 ("sensitivity(level 3)", lambda: L.lexls_lse_sensitivity(s._h, None, C.c_int32(3), C.c_double(1e-8), C.c_double(1e-12)))]:
    # (lexls_lse_solve is not in this list: it launches nothing while the solution of the current factor is cached)
    t = timed(fn, s.synchronize, 20)
    out[name] = dict(ms=1e3 * t, factor_read_GBs=fac_bytes / t / 1e9)

n2, dims2 = 40, [6] * 5
lod2 = P.lse_batch_fast(5, 1024, n2, dims2)
s2 = lexls_amd.BatchedLexLSE(1024, n2, dims2)
s2.setProblem(lod2)
s2.factorize_solve(True)
t = timed(lambda: L.lexls_lse_solve_least_norm(s2._h), s2.synchronize, 5)
out["least_norm_givens(1024 x n=40, 5x6)"] = dict(ms=1e3 * t)

# ---- configs[4]: lock-step batched LSI, warm-started ----------------------------------------------------------------------
n, dims, batch = 40, [12] * 5, int(os.environ.get("LSI_BATCH", "1024"))
base = lexlsi.pack_batch(n, [P.lsi_problem(20260500 + b, n, dims) for b in range(batch)])  # problem generation and flattening are not timed
lexlsi.lsi_batch_solve(n, base)  # warm-up: library load, first launches
t0 = time.perf_counter()
cold = lexlsi.lsi_batch_solve(n, base)
t_cold = time.perf_counter() - t0
pert = lexlsi.pack_batch(n, [P.lsi_problem(20260500 + b, n, dims, perturb=0.05) for b in range(batch)])
guess = np.where(cold["active"] == 3, 0, cold["active"]).astype(np.uint8)
t0 = time.perf_counter()
warm = lexlsi.lsi_batch_solve(n, pert, active_guess=guess, x0=cold["x"])
t_warm = time.perf_counter() - t0
# BASELINE.md C5 asks for ~30 warm-started iterations per problem: perturbation 0.9 * N(0,1) of every right-hand side (tuned with the oracle)
pert30 = lexlsi.pack_batch(n, [P.lsi_problem(20260500 + b, n, dims, perturb=0.9) for b in range(batch)])
t0 = time.perf_counter()
warm30 = lexlsi.lsi_batch_solve(n, pert30, active_guess=guess, x0=cold["x"])
t_warm30 = time.perf_counter() - t0
f30 = np.array([i["factorizations"] for i in warm30["info"]])
# the serving form: ONE batch object (device buffers, pinned blocks, streams, worker pool made once) fed the same three problem sets
srv = lexlsi.LsiBatch(n, base.dims, base.types, batch)
srv.run(base)
t_srv = {}
for name, pk, kw in (("cold", base, {}), ("warm", pert, dict(active_guess=guess, x0=cold["x"])), ("warm_30", pert30, dict(active_guess=guess, x0=cold["x"]))):
    t0 = time.perf_counter()
    r = srv.run(pk, **kw)
    t_srv[name] = time.perf_counter() - t0
    ref = {"cold": cold, "warm": warm, "warm_30": warm30}[name]
    assert np.array_equal(r["x"], ref["x"]) and r["info"] == ref["info"], f"persistent batch differs from the one-shot call ({name})"
srv.close()
# spot check against the oracle-backed stand-alone driver (first 24 instances of the cold batch): same counters, same x
from oracle import oracle_ctypes as _O
_chk = 0
for b in range(min(24, batch)):
    o = _O.lsi_run(n, P.lsi_problem(20260500 + b, n, dims))
    assert cold["info"][b] == o["info"] and np.array_equal(cold["x"][b], o["x"]), f"lock-step instance {b} differs from its stand-alone solve"
    _chk += 1
fc = np.array([i["factorizations"] for i in cold["info"]])
fw = np.array([i["factorizations"] for i in warm["info"]])
out["config4_lsi_lockstep"] = dict(batch=batch, instances_checked_against_oracle=_chk, cold=dict(seconds=t_cold, mean_factorizations=float(fc.mean()), max=int(fc.max()), rounds=cold["rounds"],
                                                         solved=int(sum(i["status"] == 0 for i in cold["info"]))),
                                   warm=dict(seconds=t_warm, mean_factorizations=float(fw.mean()), max=int(fw.max()), rounds=warm["rounds"],
                                             solved=int(sum(i["status"] == 0 for i in warm["info"])), factorizations_per_s=float(fw.sum() / t_warm)),
                                   warm_30=dict(seconds=t_warm30, mean_factorizations=float(f30.mean()), max=int(f30.max()), rounds=warm30["rounds"],
                                                solved=int(sum(i["status"] == 0 for i in warm30["info"])), factorizations_per_s=float(f30.sum() / t_warm30),
                                                perturbation=0.9),
                                   persistent_batch_seconds=t_srv,
                                   note="wall time of lexls_lsi_batch_solve on pre-packed problems: host active-set driver (worker pool) + one block copy each way per stage + device-side gather of the active rows + kernels")
print(json.dumps(out, indent=1))
