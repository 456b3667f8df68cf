"""A/B aid: held-plan merge_path_flat on a gather-bound (C2) and two L1-local (FEM-like band, fp32 / fp64)
matrices; run once per library (LOOPS_AMD_LIB) and compare.  Prints us per SpMV (median of 50) + bit-exactness."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from loops_amd import generate as G, spmv as S
from oracle import oracle as O

def ev(fn, iters=50, warm=5):
    for _ in range(warm): fn()
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(iters)]
    for a, b in evs:
        a.record(); fn(); b.record()
    torch.cuda.synchronize()
    return float(np.median([a.elapsed_time(b) for a, b in evs])) * 1e3

rows = cols = 1 << 20
cases = {"c2": (G.powerlaw_degrees(rows, 1 << 24), None), "band64": (np.full(rows, 16, np.int64), 64),
         "pl_runs": (G.powerlaw_degrees(rows, 1 << 24), -1), "pl_band64": (G.powerlaw_degrees(rows, 1 << 24), 64),
         "u16_runs": (np.full(rows, 16, np.int64), -1)}
only = sys.argv[1:]
if only:
    cases = {k: v for k, v in cases.items() if k in only}
for name, (deg, window) in cases.items():
    off, idx, val = G.csr_from_degrees(deg, cols, 1, 0, True, window)
    xh = G.uniform_distribution_int(cols)
    for dt in (np.float32, np.float64):
        if dt is np.float64 and name != "band64": continue
        v, xx = val.astype(dt), xh.astype(dt)
        ref = O.spmv_f32(off, idx, v, xx, omp=True) if dt is np.float32 else O.spmv_f64(off, idx, v, xx)
        csr = S.CSR.from_numpy(rows, cols, off, idx, v)
        x = torch.from_numpy(xx).cuda(); y = torch.empty(rows, device="cuda", dtype=x.dtype)
        out = []
        for tile in ("256x8", "128x7", "256x7", "512x8", "256x16"):
            plan = S.MergePathPlan(csr, tile)
            us = ev(lambda: S.merge_path_flat(csr, x, y, plan=plan))
            out.append(f"{tile} {us:7.1f}us self={int(plan.self_complete)} ok={bool(np.array_equal(y.cpu().numpy(), ref))}")
        for sched in ("work_oriented", "group_mapped"):
            if dt is np.float64: continue
            us = ev(lambda: S.spmv(sched, csr, x, y), iters=20)
            out.append(f"{sched} {us:7.1f}us ok={bool(np.array_equal(y.cpu().numpy(), ref))}")
        print(f"{name:8s} {np.dtype(dt).name}: " + " | ".join(out), flush=True)
