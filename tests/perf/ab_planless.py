import os, sys
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from loops_amd import generate as G, spmv as S
from oracle import oracle as O
def batch(fn, iters=100, warm=10):
    for _ in range(warm): fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / iters * 1e3
lr = int(os.environ.get("LOG_ROWS", "21"))
rows = cols = 1 << lr
deg = G.powerlaw_degrees(rows, 1 << (lr + 4))
xi = G.uniform_distribution_int(cols); x = torch.from_numpy(xi).cuda(); y = torch.empty(rows, device="cuda")
for tag, window in (("scattered", None), ("band8192", 8192), ("runs", -1)):
    off, idx, val = G.csr_from_degrees(deg, cols, 1, 0, True, window)
    csr = S.CSR.from_numpy(rows, cols, off, idx, val)
    ref = O.spmv_f32(off, idx, val, xi, omp=True)
    us = batch(lambda: S.spmv("merge_path_flat", csr, x, y))
    plan = S.MergePathPlan(csr, "512x8")
    us_plain = batch(lambda: S.merge_path_flat(csr, x, y, plan=plan, variant=0))
    S.spmv("merge_path_flat", csr, x, y)
    print(f"{tag:10s} plan-less call {us:7.1f} us (held plain plan {us_plain:7.1f}) exact={bool(np.array_equal(y.cpu().numpy(), ref))}", flush=True)
    for dt in (np.float64,):
        c64 = S.CSR.from_numpy(rows, cols, off, idx, val.astype(dt)); x64 = torch.from_numpy(xi.astype(dt)).cuda()
        y64 = S.spmv("merge_path_flat", c64, x64)
        print(f"{tag:10s} f64 plan-less {batch(lambda: S.spmv('merge_path_flat', c64, x64, y64)):7.1f} us exact={bool(np.array_equal(y64.cpu().numpy(), ref.astype(dt)))}", flush=True)
