"""Which kernels a lock-step LexLSI batch of a given shape runs (use under rocprofv3 --kernel-trace --stats).  usage: lsi_shape_probe.py n d0,d1,.. [simple_bounds 0/1]"""
import sys, os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lexls_amd import lexlsi, problems as P
n = int(sys.argv[1]); dims = [int(v) for v in sys.argv[2].split(",")]; sb = bool(int(sys.argv[3])) if len(sys.argv) > 3 else True
r = lexlsi.lsi_batch_solve(n, [P.lsi_problem(31000 + b, n, dims, simple_bounds=sb) for b in range(64)])
print("solved", sum(i["status"] == 0 for i in r["info"]), "of", len(r["info"]))
