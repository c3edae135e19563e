"""Generates tests/golden/lse_golden.npz from the CPU oracle (run from the repo root: python tests/golden/make_golden.py).

The reference itself cannot be built here (Eigen absent), so these vectors are produced by
oracle/ — they pin the oracle against regressions and give the GPU tests committed expected outputs;
the vectors that come from the reference are tests/golden/test_01.dat (copied data file,
/root/reference/tests/test_01.dat)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from lexls_amd import problems as P  # noqa: E402
from oracle import oracle_ctypes as oc  # noqa: E402

out = {}


def add(name, lod, dims, n):
    m = sum(dims)
    types = np.stack([1 + (P.uniform(1234 + b, m) * 3).astype(np.uint8) for b in range(lod.shape[0])])
    r = oc.lse_run(lod, dims, n, sens_obj=len(dims) - 1, ctr_type=types)
    out[f"{name}_lod"], out[f"{name}_dims"], out[f"{name}_nvar"], out[f"{name}_types"] = lod, np.array(dims), np.array(n), types
    for key in ("x", "factor", "hh", "perm", "rank", "v", "lam", "sens"):
        out[f"{name}_{key}"] = r[key]


add("ik", P.lse_batch(20260100, 6, 40, [12] * 5), [12] * 5, 40)
add("rankdef", np.stack([P.rank_deficient_problem(100 + b, 15, [5] * 4, [3] * 4) for b in range(6)]), [5] * 4, 15)
np.savez_compressed(os.path.join(ROOT, "tests", "golden", "lse_golden.npz"), **out)
print("wrote", {k: v.shape for k, v in out.items() if k.endswith("_x")})
