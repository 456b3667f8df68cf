"""The plan-less (asynchronous, one-shot) entries of the three named schedules on the three C3 stand-ins (indochina-2004's shape;
x = 28 MB): group_mapped / work_oriented / merge_path_flat through loops_spmv_csr_f32 -- from an x of 6 MB on the device
samples the columns and gathers in phases where they are scattered (round 4).  us per call + bit-exactness vs the oracle."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from loops_amd import generate as G, spmv as S
from oracle import oracle as O

def batch(fn, iters=15, warm=3):
    for _ in range(warm): fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / iters * 1e3

rows = cols = 7_414_866
nnz = 194_109_311
deg = G.powerlaw_degrees(rows, nnz)
xh = G.uniform_distribution_int(cols)
x = torch.from_numpy(xh).cuda()
for tag, window in (("uniform", None), ("band_65536", 65536), ("host_blocked", G.HOST_BLOCKED)):
    off, idx, val = G.csr_from_degrees(deg, cols, 1, 0, True, window, hosts=G.host_blocks(cols) if window == G.HOST_BLOCKED else None)
    csr = S.CSR.from_numpy(rows, cols, off, idx, val)
    ref = O.spmv_f32(off, idx, val, xh, omp=True)
    y = torch.empty(rows, device="cuda")
    out = []
    for sched in ("group_mapped", "work_oriented", "merge_path_flat"):
        us = batch(lambda: S.spmv(sched, csr, x, y))
        out.append(f"{sched} {us:7.1f} exact={bool(np.array_equal(y.cpu().numpy(), ref))}")
    print(f"C3 stand-in {tag}: " + " | ".join(out), flush=True)
    del csr, off, idx, val
