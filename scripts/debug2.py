import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from loops_amd import generate as G, spmv as S, _lib
from oracle import oracle as O
rows, nnz, cap = 64, 600, 64
deg = G.powerlaw_degrees(rows, nnz, cap=cap)
off, idx, val = G.powerlaw_csr(rows, rows, nnz, degrees=deg)
x = G.uniform_distribution_int(rows)
print("off", off[:8], "prods", (val[:8]*x[idx[:8]]))
ts, owner, row, vis = O.merge_path_assign(off, 256, 8)
print("oracle thread starts", ts[:6].tolist())
csr = S.CSR.from_numpy(rows, rows, off, idx, val)
plan = S.MergePathPlan(csr)
y = S.merge_path_flat(csr, torch.from_numpy(x).cuda(), plan=plan); torch.cuda.synchronize()
print(y[:8].cpu().numpy(), O.spmv_f32(off, idx, val, x)[:8])
