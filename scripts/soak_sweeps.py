"""Soak (not part of the suite): the seeded sweep tests of tests/test_gpu_random_sweep.py with fresh seeds (chunk numbers beyond the suite's),
for a time budget.  usage: python scripts/soak_sweeps.py [seconds]"""
import os, sys, time, inspect
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import lexls_amd as hip
from oracle import oracle_ctypes as oracle
import test_gpu_random_sweep as T

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
fns = [(n, f) for n, f in inspect.getmembers(T, inspect.isfunction) if n.startswith("test_random")]
t0, chunk, runs = time.time(), 1000, {}
while time.time() - t0 < budget:
    for name, f in fns:
        try:
            f(hip, oracle, chunk)
        except AssertionError as e:  # the suite's "this chunk saw kernel X" checks hold for the suite's seeds, not for every seed
            if "seen" in str(e) or str(e).strip().startswith("{") or "len(kernels)" in str(e):
                pass
            else:
                raise
        runs[name] = runs.get(name, 0) + 1
        if time.time() - t0 >= budget:
            break
    chunk += 1
print(f"soak ok in {time.time() - t0:.0f} s: " + ", ".join(f"{k} x{v}" for k, v in runs.items()))
