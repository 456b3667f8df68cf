import sys, time
sys.path.insert(0, "/root/repo")
import numpy as np, torch
from loops_amd import generate as G, spmv as S
rows = cols = 1 << 20
off, idx, val = G.powerlaw_csr(rows, cols, 1 << 24)
csr = S.CSR.from_numpy(rows, cols, off, idx, val)
ts = []
for i in range(8):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    p = S.RowBandPlan(csr)
    torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
    p.close()
print("build wall ms:", [round(t, 3) for t in ts])
