"""BASELINE configs[0] (the reference's tests/test_01.dat: n = 88, levels [33 simple bounds, 3, 2, 97]) as ONE LexLSI problem through the C ABI
(lexls_lsi_solve_dat: parsing the file is part of both timings): wall time per solve, factorizations, and the oracle-backed driver on one host
core beside it.  usage: python scripts/time_config0.py [repeats]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from lexls_amd import lexlsi
from oracle import oracle_ctypes as oracle

DAT = os.path.join(ROOT, "tests", "golden", "test_01.dat")
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
r = lexlsi.lsi_solve_dat(DAT, 88)
assert np.abs(r["x"] - r["solution"]).max() < 1e-10
t, tc = [], []
for _ in range(reps):
    t0 = time.perf_counter()
    r = lexlsi.lsi_solve_dat(DAT, 88)
    t.append(time.perf_counter() - t0)
for _ in range(5):
    t0 = time.perf_counter()
    o = oracle.lsi_run_dat(DAT)
    tc.append(time.perf_counter() - t0)
print(f"config0: {r['info']['factorizations']} factorizations; device path {1e3 * min(t):.2f} ms (median {1e3 * float(np.median(t)):.2f}); "
      f"oracle-backed driver on one host core {1e3 * min(tc):.2f} ms")
