"""Lock-step LexLSI batch of problems with the shape of BASELINE configs[0] (n = 88, levels [33 simple bounds, 3, 2, 97]) — beyond the
one-wavefront kernels: generic kernel + host-side active-set logic.  usage: python scripts/time_lsi_wide.py [batch]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from lexls_amd import lexlsi, problems as P
batch = int(sys.argv[1]) if len(sys.argv) > 1 else 256
n, dims = 88, [33, 3, 2, 97]
if len(sys.argv) > 2 and sys.argv[2] == "n47":  # 48 columns: beyond the 41-column register-resident instantiation
    n, dims = 47, [12, 12, 12, 12, 12]
if len(sys.argv) > 2 and sys.argv[2] == "n55":  # the 64-column shapes
    n, dims = 55, [12, 16, 14, 16]
if len(sys.argv) > 2 and sys.argv[2] == "n12":  # one slot of the four-per-wavefront kernel
    n, dims = 12, [6, 4, 4, 4]
if len(sys.argv) > 2 and sys.argv[2] == "deep":  # IK-sized problem with eight levels: 12 simple bounds + 7 x 12 rows (deep hierarchy: left-looking kernels)
    n, dims = 40, [12] * 8
problems = [P.lsi_problem(9000 + b, n, dims, simple_bounds=True) for b in range(batch)]
pk = lexlsi.pack_batch(n, problems)
b = lexlsi.LsiBatch(n, pk.dims, pk.types, pk.batch)
os.environ["LEXLS_LSI_TIMING"] = "1"
for rep in range(3):
    t0 = time.perf_counter()
    r = b.run(pk)
    dt = time.perf_counter() - t0
    f = np.array([i["factorizations"] for i in r["info"]])
    print(f"batch {batch}: {dt * 1e3:.1f} ms, factorizations mean {f.mean():.1f} max {f.max()}, {f.sum() / dt:.3e} fact/s, status ok {sum(i['status'] == 0 for i in r['info'])}/{batch}, stats {b.stats() if hasattr(b, 'stats') else ''}")
