import sys; sys.path.insert(0,'.')
import numpy as np
import lexls_amd
from lexls_amd import problems as P
from oracle import oracle_ctypes as oc
n,dims=40,[12]*5
lod=P.lse_batch(20260100,2,n,dims)
ref=oc.lse_run(lod,dims,n)
s=lexls_amd.BatchedLexLSE(2,n,dims); s.setProblem(lod); s.factorize_solve(True)
f=s.get_lexqr(); 
d=(f[0]!=ref['factor'][0])  # (41, 60): [col][row]
print('x equal', np.array_equal(s.get_x(), ref['x']), np.abs(s.get_x()-ref['x']).max())
print('mismatch per col (position):', d.sum(axis=1).tolist())
print('mismatch per row:', d.sum(axis=0).tolist())
rows,cols=np.where(d.T)
print(list(zip(rows[:20].tolist(), cols[:20].tolist())))
i,j=rows[0],cols[0]
print(f[0,j,i], ref['factor'][0,j,i])
