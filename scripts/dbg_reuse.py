import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, lexls_amd as hip
import test_gpu_prefix_reuse as T
seed = 1
rng = np.random.default_rng(seed)
b1, b2, K = [], [], []
for b in range(97):
    dims = rng.integers(5, 13, size=5)
    blocks = T.random_blocks(1000 * seed + b, dims)
    k = int(rng.integers(0, 6))
    b1.append(blocks); b2.append(T.change_level(blocks, k, 7000 * seed + b, int(rng.integers(0, 3)))); K.append(k)
for pb in [22]:
    lod1, dims1 = T.pack([b1[pb]]); lod2, dims2 = T.pack([b2[pb]])
    print("problem", pb, "K", K[pb], "dims1", dims1, "dims2", dims2)
    s = hip.BatchedLexLSE(1, 40, [12] * 5); s.set_kernel_policy(2); s.set_prefix_reuse(True)
    s.setObjDim(dims1); s.setProblem(lod1); s.factorize_solve(True)
    f1 = s.get_lexqr().copy(); p1 = s.get_column_permutations().copy(); print("ranks1", s.getRanks())
    s.setObjDim(dims2); s.setProblem(lod2); s.set_resume_levels(np.array([K[pb]], np.int32)); s.factorize_solve(True)
    f2 = s.get_lexqr(); print("ranks2", s.getRanks())
    r = hip.BatchedLexLSE(1, 40, [12] * 5); r.set_kernel_policy(2); r.setObjDim(dims2); r.setProblem(lod2); r.factorize_solve(True)
    fr = r.get_lexqr()
    m = int(dims2.sum())
    d = np.argwhere(f2[0][:, :m] != fr[0][:, :m])
    print("differing (column position, row):", d.tolist())
    for c, i in d[:10]: print(c, i, f2[0][c, i], fr[0][c, i], "first run had", f1[0][c, i])
    print("perm equal", np.array_equal(p1, s.get_column_permutations()), np.array_equal(s.get_column_permutations(), r.get_column_permutations()))
