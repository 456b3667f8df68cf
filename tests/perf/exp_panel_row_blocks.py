"""Experiment: the panel-binned copy of a LARGE matrix built as R independent copies over contiguous row blocks, run back to
back on one stream (y ranges are disjoint), against the one copy over all rows.  Question: does the whole-matrix copy of C5
(2 GB of products between its two kernels, 512 KB between a panel's store runs) pay for its size?
usage: exp_panel_row_blocks.py [c5|c5_shard|c3_uniform] [R ...] [--hw=H]"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import numpy as np, torch
from loops_amd import generate as G, spmv as S, partition as P
from bench_panel_cases import CASES, batch_ms  # noqa: E402

cases = dict(CASES, c5=(1 << 24, 1 << 24, 1 << 29, None))
name = next((a for a in sys.argv[1:] if a in cases), "c5")
Rs = [int(a) for a in sys.argv[1:] if a.isdigit()] or [1, 2, 4, 8, 16]
HW = next((int(a[5:]) for a in sys.argv[1:] if a.startswith("--hw=")), 0)   # force the sub-band height of every copy (0 = automatic)
rows, cols, nnz, window = cases[name]
deg = G.powerlaw_degrees(rows, nnz, cap=min(1 << 14, cols))
off, idx, val = G.csr_from_degrees(deg, cols, 1, 0, True, window)
x = torch.from_numpy(G.uniform_distribution_int(cols)).cuda()
y = torch.empty(rows, device="cuda")
ref = None
out = {"case": name, "rows": rows, "nnz": nnz, "runs": []}
for R in Rs:
    bounds = P.row_ranges(off, R) if R > 1 else np.array([0, rows])
    plans, ys = [], []
    for r in range(R):
        b, e = int(bounds[r]), int(bounds[r + 1])
        o, i, v = P.slice_csr(off, idx, val, b, e)
        c = S.CSR.from_numpy(e - b, cols, o, i, v)
        plans.append((c, S.PanelBinnedPlan(c, HW)))
        ys.append(y[b:e])
    def run():
        for (c, p), yy in zip(plans, ys):
            p.spmv(x, yy)
    def stage(s):
        for (c, p), yy in zip(plans, ys):
            p.spmv_stage(s, x, yy)
    y.zero_()
    run()
    torch.cuda.synchronize()
    if ref is None:
        ref = y.clone()
    eq = bool(torch.equal(y, ref))
    t = batch_ms(run, iters=10)
    ta = batch_ms(lambda: stage(0), iters=10)
    tb = batch_ms(lambda: stage(1), iters=10)
    p0 = plans[0][1]
    row = {"row_blocks": R, "ms": round(t, 4), "products_ms": round(ta, 4), "reduce_ms": round(tb, 4), "equal_to_one_copy": eq,
           "W": p0.W, "Hw": p0.Hw, "panels": p0.num_panels, "subbands_first_block": p0.num_subbands, "compact": bool(p0.compact)}
    out["runs"].append(row)
    print(json.dumps(row), file=sys.stderr, flush=True)
    for c, p in plans:
        p.close()
    del plans, ys
print(json.dumps(out))
