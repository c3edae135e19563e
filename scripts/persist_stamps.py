import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, lexls_amd as hip
from lexls_amd import problems as P
n, dims = 512, [256] * 4
lod = P.lse_batch(20260001, 1, n, dims)
s = hip.BatchedLexLSE(1, n, dims); s.setProblem(lod)
for _ in range(3): s.factorize()
s.synchronize()
w = s.getWorkspace()[0]
names = ["flush pending", "wait for the records", "POLLS x1000", "read column + norms", "publish column", "dot + wave sums", "scalars", "maps + pivot row + norms", "barrier C", "publish record", "column update", "COLUMN READS x1000"]
for lvl in range(2):
    v = w[16 * lvl: 16 * lvl + 12]
    print("level", lvl, "torn records / torn value granules seen by workgroup 1 (tag present, checksum wrong):", int(w[16 * lvl + 12]), int(w[16 * lvl + 13]))
    print("level", lvl, "cycles per pivot:", {nm: round(x / 256 * (1000 if "x1000" in nm else 1)) for nm, x in zip(names, v)}, "total/pivot", round((v.sum() - v[2] - v[11]) / 256))
# placement of the workgroups (HW_ID: cu_id bits 11:8, sh 12, se 15:13) and each one's wait for the records, level 0
import collections
G = 0
rows = []
for t in range(300):
    hw, wait = w[64 + 2 * t], w[64 + 2 * t + 1]
    if hw == 0 and wait == 0: break
    hw = int(hw); rows.append(((hw >> 13) & 7, (hw >> 12) & 1, (hw >> 8) & 15, wait / 256))
per_cu = collections.Counter((r[0], r[1], r[2]) for r in rows)
print("workgroups:", len(rows), "distinct CUs:", len(per_cu), "workgroups per CU histogram:", sorted(collections.Counter(per_cu.values()).items()))
waits = sorted(r[3] for r in rows)
if waits: print("wait for records, cycles per pivot: min %.0f median %.0f max %.0f" % (waits[0], waits[len(waits) // 2], waits[-1]))
