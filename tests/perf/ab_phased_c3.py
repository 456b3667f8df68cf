"""Phased against plain merge_path_flat on the three C3 stand-ins (indochina-2004's shape: uniform columns, 65 536-wide band,
host-blocked): what phasing costs where the gathers are local and what it buys where they are not.  us per whole step."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from loops_amd import generate as G, spmv as S, _lib

def batch(fn, iters=20, warm=3):
    for _ in range(warm): fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / iters * 1e3

rows = cols = 7_414_866
nnz = 194_109_311
deg = G.powerlaw_degrees(rows, nnz)
x = torch.from_numpy(G.uniform_distribution_int(cols)).cuda()
for tag, window in (("uniform", None), ("band_65536", 65536), ("host_blocked", G.HOST_BLOCKED)):
    off, idx, val = G.csr_from_degrees(deg, cols, 1, 0, True, window, hosts=G.host_blocks(cols) if window == G.HOST_BLOCKED else None)
    csr = S.CSR.from_numpy(rows, cols, off, idx, val)
    y = torch.empty(rows, device="cuda"); y0 = torch.empty(rows, device="cuda")
    out = []
    for tile in ("256x8", "512x8", "256x16"):
        plan = S.MergePathPlan(csr, tile)
        S.merge_path_flat(csr, x, y0, plan=plan, variant=0)
        for v in ((0,) if tile == "256x8" else (0, _lib.VARIANT_PHASED)):
            s = min(batch(lambda: S.merge_path_flat(csr, x, y, plan=plan, variant=v)) for _ in range(2))
            out.append(f"{tile}{'+phased' if v else ''} {s:7.1f} eq={bool(torch.equal(y, y0))}")
        plan.close()
    print(f"C3 stand-in {tag}: guess_scattered={S.columns_look_scattered(csr)} | " + " | ".join(out), flush=True)
    del csr, off, idx, val
