import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from scripts.dbg_mfma import report, N, DIMS
from lexls_amd import problems as P
pol = int(sys.argv[1]) if len(sys.argv) > 1 else 7
report("two-level", P.lse_batch(5, 8, N, [12, 12]), [12, 12], N, pol)
report("ik-5", P.lse_batch(2, 5, N, DIMS), DIMS, N, pol)
