"""Where the row-band layout loses: stage times, chunk / piece counts, padding and hub rows for the inputs `sweep_structures.py` /
`bench_panel.py` show it behind panel-binned on.  usage: diag_rowband_structures.py [runs|rmat_none|rmat_degree|<a case of
bench_panel_cases.CASES> ...] [--cfg=H,target_chunks ...]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import json
import numpy as np, torch
from loops_amd import generate as G, spmv as S
from bench_panel_cases import CASES, batch_ms  # noqa: E402

N = 1 << 24
def make(name):
    if name == "runs":
        return G.csr_from_degrees(G.powerlaw_degrees(1 << 20, N), 1 << 20, 1, 0, True, -1)
    if name in CASES:
        rows, cols, nnz, window = CASES[name]
        deg = G.powerlaw_degrees(rows, nnz, cap=min(1 << 14, cols)) if name != "short_rows_8M" else np.full(rows, 2, np.int64)
        return G.csr_from_degrees(deg, cols, 1, 0, True, window, hosts=G.host_blocks(cols) if window == G.HOST_BLOCKED else None)
    return G.rmat_csr(20, 16, relabel=name[5:])

want = [a for a in sys.argv[1:] if not a.startswith("-")] or ["runs", "rmat_none", "rmat_degree"]
for name in want:
    off, idx, val = make(name)
    rows = off.size - 1; cols = CASES[name][1] if name in CASES else 1 << 20; nnz = int(off[-1])
    csr = S.CSR.from_numpy(rows, cols, off, idx, val)
    x = torch.from_numpy(G.uniform_distribution_int(cols)).cuda()
    y0, y1 = torch.empty(rows, device="cuda"), torch.empty(rows, device="cuda")
    mp = S.MergePathPlan(csr, "512x8")
    S.merge_path_flat(csr, x, y0, plan=mp)
    for H, target in [(0, 0)] + [(int(a.split("=")[1].split(",")[0]), int(a.split("=")[1].split(",")[1])) for a in sys.argv[1:] if a.startswith("--cfg=")]:
        rb = S.RowBandPlan(csr, H, target)
        rb.spmv(x, y1)
        arr = rb.arrays()
        chunks, multi = arr[5], arr[6]
        band_steps = np.bincount(chunks[:, 0], weights=(chunks[:, 2] - chunks[:, 1]), minlength=rb.num_bands) if chunks.shape[1] >= 3 else None
        row = {"case": name, "H": rb.H, "bands": rb.num_bands, "chunks": rb.num_chunks, "cut_bands": rb.num_multi, "partials": rb.num_partials,
               "max_pieces_per_band": int(multi[:, 2].max()) if len(multi) else 1, "padding_pct": round(100.0 * (rb.padded - nnz) / nnz, 2),
               "us": round(batch_ms(lambda: rb.spmv(x, y1)) * 1e3, 1), "accumulate_us": round(batch_ms(lambda: rb.spmv_stage(0, x, y1)) * 1e3, 1),
               "combine_us": round(batch_ms(lambda: rb.spmv_stage(1, x, y1)) * 1e3, 1) if rb.num_multi else 0.0,
               "equal": bool(torch.equal(y0, y1))}
        if band_steps is not None:
            row["band_steps_max_over_mean"] = round(float(band_steps.max() / band_steps.mean()), 2)
        hubs = arr[7]
        row["hub_rows"] = int(hubs.reshape(rb.num_bands, -1)[:, 0].sum()) if hubs.size else 0
        print(json.dumps(row), flush=True)
        rb.close()
