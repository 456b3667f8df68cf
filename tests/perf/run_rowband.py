"""Driver for profiler passes over the row-band product: usage run_rowband.py <case> [H] [target_chunks] [iterations]
(cases of bench_panel_cases.py; kernel shape through LOOPS_ROWBAND_CFG)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import numpy as np, torch
from loops_amd import generate as G, spmv as S
from bench_panel_cases import CASES

name = sys.argv[1] if len(sys.argv) > 1 else "c2"
H = int(sys.argv[2]) if len(sys.argv) > 2 else 0
chunks = int(sys.argv[3]) if len(sys.argv) > 3 else 0
iters = int(sys.argv[4]) if len(sys.argv) > 4 else 30
rows, cols, nnz, window = CASES[name]
deg = G.powerlaw_degrees(rows, nnz, cap=min(1 << 14, cols)) if name != "short_rows_8M" else np.full(rows, 2, np.int64)
hosts = G.host_blocks(cols) if window == G.HOST_BLOCKED else None
off, idx, val = G.csr_from_degrees(deg, cols, 1, 0, True, window, hosts=hosts)
csr = S.CSR.from_numpy(rows, cols, off, idx, val)
x = torch.from_numpy(G.uniform_distribution_int(cols)).cuda()
y = torch.empty(rows, device="cuda")
plan = S.RowBandPlan(csr, H, chunks)
for _ in range(iters):
    plan.spmv(x, y)
torch.cuda.synchronize()
print(name, "H", plan.H, "chunks", plan.num_chunks, "partials", plan.num_partials)
