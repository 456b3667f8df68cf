"""Panel-binned layout vs plain CSR vs row-band on the configurations whose x exceeds (or fills) an L2: C2, one rank's
shard of C5, the C3 stand-ins.  Times per product (back-to-back batch between one event pair), stage times of the
panel-binned kernels, bytes moved per nonzero, equality of the three results.  usage: bench_panel.py [case ...]"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from loops_amd import generate as G, spmv as S


sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from bench_panel_cases import CASES, batch_ms, build_ms  # noqa: E402

F64 = "--f64" in sys.argv   # the same cases with 8-byte values
want = [a for a in sys.argv[1:] if a in CASES] or ["c2", "c5_shard", "c3_uniform", "c3_host_blocked"]
out = {}
for name in want:
    rows, cols, nnz, window = CASES[name]
    deg = G.powerlaw_degrees(rows, nnz, cap=min(1 << 14, cols)) if name != "short_rows_8M" else np.full(rows, 2, np.int64)
    hosts = G.host_blocks(cols) if window == G.HOST_BLOCKED else None
    off, idx, val = G.csr_from_degrees(deg, cols, 1, 0, True, window, hosts=hosts)
    vb = 8 if F64 else 4
    csr = S.CSR.from_numpy(rows, cols, off, idx, val.astype(np.float64) if F64 else val)
    x = torch.from_numpy(G.uniform_distribution_int(cols)).cuda()
    x = x.double() if F64 else x
    y0, y1, y2 = (torch.empty(rows, device="cuda", dtype=x.dtype) for _ in range(3))
    abytes = nnz * (4 + vb) + (rows + 1) * 4 + rows * vb + cols * vb
    mp, b_csr = build_ms(lambda: S.MergePathPlan(csr, "512x8"))
    t_csr = batch_ms(lambda: S.merge_path_flat(csr, x, y0, plan=mp))
    if F64:   # (the row-band copy holds 4-byte values only)
        cb, b_cb, t_cb, blocks = None, float("nan"), float("inf"), 0
        y1.copy_(y0)
    else:
        cb, b_cb = build_ms(lambda: S.RowBandPlan(csr))
        cb.tune(5)
        t_cb = batch_ms(lambda: cb.spmv(x, y1))
        blocks = cb.num_bands
        cb.close()
    for hw in [int(t) for t in os.environ.get("PANEL_HW", "").split(",") if t]:  # tuning aid: explicit sub-band heights
        pv = S.PanelBinnedPlan(csr, hw, int(os.environ.get("PANEL_W", "0")))
        print(name, "W", pv.W, "Hw", hw, "subbands", pv.num_subbands, "total %.1f us  products %.1f  reduce %.1f" % (
            batch_ms(lambda: pv.spmv(x, y2)) * 1e3, batch_ms(lambda: pv.spmv_stage(0, x, y2)) * 1e3, batch_ms(lambda: pv.spmv_stage(1, x, y2)) * 1e3),
            file=sys.stderr, flush=True)
        pv.close()
    pb, b_pb = build_ms(lambda: S.PanelBinnedPlan(csr))
    t_pb = batch_ms(lambda: pb.spmv(x, y2))
    t_a = batch_ms(lambda: pb.spmv_stage(0, x, y2))
    t_b = batch_ms(lambda: pb.spmv_stage(1, x, y2))
    pb.spmv(x, y2)
    row = {"rows": rows, "cols": cols, "nnz": nnz, "dtype": "f64" if F64 else "f32", "x_MB": cols * vb >> 20, "algorithmic_bytes": abytes,
           "csr_512x8_ms": round(t_csr, 4), "row_band_ms": round(t_cb, 4), "row_bands": blocks,
           "panel_binned_ms": round(t_pb, 4), "panel_products_ms": round(t_a, 4), "panel_reduce_ms": round(t_b, 4),
           "panel": {"W": pb.W, "Hw": pb.Hw, "panels": pb.num_panels, "subbands": pb.num_subbands, "padding_items": pb.padded - nnz,
                     "chunks": pb.num_chunks, "items_per_segment": round(nnz / (pb.num_panels * pb.num_subbands), 1)},
           "frac_csr": round(abytes / t_csr / 1e6 / 8000, 4), "frac_row_band": round(abytes / t_cb / 1e6 / 8000, 4),
           "frac_panel": round(abytes / t_pb / 1e6 / 8000, 4),
           "products_GBps": round(pb.padded * (3 + 2 * vb) / t_a / 1e6, 1), "reduce_GBps": round(pb.padded * (2 + vb) / t_b / 1e6, 1),
           "equal": bool(torch.equal(y0, y1) and torch.equal(y0, y2)),
           # what a plan costs to build, and after how many products the copy has paid for itself against the held CSR plan
           "build_ms": {"csr_plan": round(b_csr, 2), "row_band": round(b_cb, 2), "panel_binned": round(b_pb, 2)},
           "products_to_amortise": {"row_band": round(b_cb / (t_csr - t_cb), 1) if t_cb < t_csr else None,
                                    "panel_binned": round(b_pb / (t_csr - t_pb), 1) if t_pb < t_csr else None}}
    out[name] = row
    print(name, json.dumps(row), file=sys.stderr, flush=True)
    pb.close()
    mp.close()
    del csr, x, y0, y1, y2, off, idx, val
print(json.dumps(out))
