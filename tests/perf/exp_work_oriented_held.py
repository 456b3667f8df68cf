import sys, os
sys.path.insert(0, "/root/repo")
import numpy as np, torch
from loops_amd import generate as G, spmv as S
from oracle import oracle as O
def ms(fn, iters=30):
    for _ in range(3): fn()
    a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize()
    return a.elapsed_time(e) / iters * 1e3
for name, cols in (("c2", 1 << 20), ("c2x8", 1 << 21), ("c2x16", 1 << 22)):
    rows = 1 << 20
    off, idx, val = G.powerlaw_csr(rows, cols, 1 << 24)
    csr = S.CSR.from_numpy(rows, cols, off, idx, val)
    xh = G.uniform_distribution_int(cols); x = torch.from_numpy(xh).cuda(); y = torch.empty(rows, device="cuda")
    want = O.spmv_f32(off, idx, val, xh, omp=True)
    plan = S.MergePathPlan(csr, "256x8")
    t = ms(lambda: S.work_oriented(csr, x, y, plan=plan))
    S.work_oriented(csr, x, y, plan=plan)
    print(name, "work_oriented held plan %.1f us exact=%s" % (t, bool(np.array_equal(y.cpu().numpy(), want))), flush=True)
