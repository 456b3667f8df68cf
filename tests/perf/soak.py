"""Soak: the tuned kernels launched back to back for a fixed wall time on C2 and on a self-completing band matrix; every
result compared ON THE GPU with the first one (bit-equal) -- races / ordering bugs show up as a mismatch count > 0.
usage: python tests/perf/soak.py [seconds per case, default 20]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from loops_amd import generate as G, spmv as S
from oracle import oracle as O

secs = float(sys.argv[1]) if len(sys.argv) > 1 else 20.0
rows = cols = 1 << 20
cases = {"c2": (G.powerlaw_degrees(rows, 1 << 24), None), "band64": (np.full(rows, 16, np.int64), 64)}
for name, (deg, window) in cases.items():
    off, idx, val = G.csr_from_degrees(deg, cols, 1, 0, True, window)
    xh = G.uniform_distribution_int(cols)
    ref = torch.from_numpy(O.spmv_f32(off, idx, val, xh, omp=True)).cuda()
    csr = S.CSR.from_numpy(rows, cols, off, idx, val)
    x = torch.from_numpy(xh).cuda()
    plans = {t: S.MergePathPlan(csr, t) for t in ("256x8", "512x8")}
    cb = S.ColumnBlockedPlan(csr)
    kernels = {f"merge_path_flat {t} (self={int(p.self_complete)})": (lambda y, p=p: S.merge_path_flat(csr, x, y, plan=p)) for t, p in plans.items()}
    kernels["work_oriented"] = lambda y: S.spmv("work_oriented", csr, x, y)
    kernels["group_mapped"] = lambda y: S.spmv("group_mapped", csr, x, y)
    kernels["column_blocked"] = lambda y: cb.spmv(x, y)
    for label, fn in kernels.items():
        y = torch.empty(rows, device="cuda")
        bad = torch.zeros((), dtype=torch.int64, device="cuda")
        rounds, t0 = 0, time.time()
        while time.time() - t0 < secs:
            for _ in range(200):
                y.fill_(float("nan"))
                fn(y)
                bad += (y != ref).any()
            rounds += 200
            torch.cuda.synchronize()
        print(f"{name:7s} {label:40s} rounds {rounds:7d} mismatching rounds {int(bad.item())}", flush=True)
