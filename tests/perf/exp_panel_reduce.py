"""Experiment driver (round 4): kernel B variants of the panel-binned layout.  One process per LOOPS_PANEL_REDUCE value (the
switch is read once); per case and sub-band height: stage times, equality with the CSR product on exactly-summable inputs,
and the worst relative error on realistic values against an f64 sum.
usage: LOOPS_PANEL_REDUCE=84 PANEL_HW=1024,2048 python tests/perf/exp_panel_reduce.py c2 c5_shard"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from loops_amd import generate as G, spmv as S
import bench_panel_cases
from bench_panel_cases import CASES, batch_ms

want = [a for a in sys.argv[1:] if a in CASES] or ["c2", "c5_shard"]
hws = [int(t) for t in os.environ.get("PANEL_HW", "0").split(",") if t]
variant = os.environ.get("LOOPS_PANEL_REDUCE", "0")
for name in want:
    rows, cols, nnz, window = CASES[name]
    deg = G.powerlaw_degrees(rows, nnz, cap=min(1 << 14, cols)) if name != "short_rows_8M" else np.full(rows, 2, np.int64)
    hosts = G.host_blocks(cols) if window == G.HOST_BLOCKED else None
    off, idx, val = G.csr_from_degrees(deg, cols, 1, 0, True, window, hosts=hosts)
    csr = S.CSR.from_numpy(rows, cols, off, idx, val)
    x = torch.from_numpy(G.uniform_distribution_int(cols)).cuda()
    mp = S.MergePathPlan(csr, "512x8")
    want_y = S.merge_path_flat(csr, x, plan=mp).clone()
    # realistic values for the accuracy figure (same structure)
    rng = np.random.default_rng(5)
    val_r = (rng.random(nnz, dtype=np.float32) + np.float32(0.5))
    xr = G.realistic_x(cols)
    prod = val_r.astype(np.float64) * xr.astype(np.float64)[idx]
    nz = np.diff(off.astype(np.int64)) > 0
    ref = np.zeros(rows)
    ref[nz] = np.add.reduceat(prod, off[:-1].astype(np.int64)[nz])
    del prod
    vr = torch.from_numpy(val_r).cuda()
    xrd = torch.from_numpy(xr).cuda()
    for hw in hws:
        try:
            cm = os.environ.get("PANEL_COMPACT", "")
            pv = S.PanelBinnedPlan(csr, hw, int(os.environ.get("PANEL_W", "0")), compact=None if cm == "" else bool(int(cm)))
        except Exception as e:  # noqa: BLE001
            print(name, "variant", variant, "Hw", hw, "FAILED", e, flush=True)
            continue
        y = torch.empty(rows, device="cuda")
        t = batch_ms(lambda: pv.spmv(x, y)) * 1e3
        ta = batch_ms(lambda: pv.spmv_stage(0, x, y)) * 1e3
        tb = batch_ms(lambda: pv.spmv_stage(1, x, y)) * 1e3
        pv.spmv(x, y)
        equal = bool(torch.equal(y, want_y))
        pv.refresh_values(vr)
        yr = pv.spmv(xrd).cpu().numpy().astype(np.float64)
        rel = np.abs(yr - ref)[nz] / np.abs(ref[nz])
        again = all(bool(torch.equal(pv.spmv(xrd), torch.from_numpy(yr.astype(np.float32)).cuda())) for _ in range(3))
        pv.refresh_values(csr.values)
        print(json.dumps({"case": name, "variant": variant, "W": pv.W, "Hw": pv.Hw, "subbands": pv.num_subbands, "compact": pv.compact, "runs_over_nnz": round(pv.runs / nnz, 4), "total_us": round(t, 1),
                          "products_us": round(ta, 1), "reduce_us": round(tb, 1), "equal": equal, "max_rel_err": float(rel.max()),
                          "reproducible": again}), flush=True)
        pv.close()
    mp.close()
    del csr, x, off, idx, val
