"""Driver for PMC passes over the planned merge_path_flat SpMV on a matrix with column locality
(2^20 rows x 16 nonzeros, columns in a 64-wide band): `rocprofv3 --pmc ... -- python scripts/pmc_local.py`."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from loops_amd import generate as G, spmv as S
rows = cols = 1 << 20
off, idx, val = G.csr_from_degrees(np.full(rows, 16, np.int64), cols, 1, 0, True, 64)
csr = S.CSR.from_numpy(rows, cols, off, idx, val)
x = torch.from_numpy(G.uniform_distribution_int(cols)).cuda()
y = torch.empty(rows, device="cuda")
plan = S.MergePathPlan(csr)
for _ in range(int(sys.argv[1]) if len(sys.argv) > 1 else 30):
    S.merge_path_flat(csr, x, y, plan=plan)
torch.cuda.synchronize()
