"""Experiment (prototype, torch ops for the split): A = A_dense + A_sparse by how many of a band's nonzeros share a 128-byte line
of x -- A_dense (lines with at least T nonzeros of the band) through the row-band copy, A_sparse through the panel-binned copy --
against either copy over all of A.  usage: exp_hybrid_split.py [case ...] [--T=2,3,4] [--H=16384]"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import numpy as np, torch
from loops_amd import generate as G, spmv as S
from bench_panel_cases import CASES, batch_ms  # noqa: E402

N = 1 << 24
def make(name):
    if name in CASES:
        rows, cols, nnz, window = CASES[name]
        deg = G.powerlaw_degrees(rows, nnz, cap=min(1 << 14, cols)) if name != "short_rows_8M" else np.full(rows, 2, np.int64)
        off, idx, val = G.csr_from_degrees(deg, cols, 1, 0, True, window, hosts=G.host_blocks(cols) if window == G.HOST_BLOCKED else None)
        return off, idx, val, cols
    off, idx, val = G.rmat_csr(20, 16, relabel=name[5:])
    return off, idx, val, 1 << 20

Ts = [int(t) for a in sys.argv[1:] if a.startswith("--T=") for t in a[4:].split(",")] or [2, 3, 4]
H = next((int(a[4:]) for a in sys.argv[1:] if a.startswith("--H=")), 16384)
for name in [a for a in sys.argv[1:] if not a.startswith("-")] or ["c3_host_blocked"]:
    off, idx, val, cols = make(name)
    rows, nnz = off.size - 1, int(off[-1])
    csr = S.CSR.from_numpy(rows, cols, off, idx, val)
    x = torch.from_numpy(G.uniform_distribution_int(cols)).cuda()
    y0, y1, y2 = (torch.empty(rows, device="cuda") for _ in range(3))
    mp = S.MergePathPlan(csr, "512x8")
    S.merge_path_flat(csr, x, y0, plan=mp)
    t_csr = batch_ms(lambda: S.merge_path_flat(csr, x, y0, plan=mp), iters=10)
    rb = S.RowBandPlan(csr, H); rb.tune(3)
    t_rb = batch_ms(lambda: rb.spmv(x, y1), iters=10); rb.close()
    pb = S.PanelBinnedPlan(csr)
    t_pb = batch_ms(lambda: pb.spmv(x, y1), iters=10); pb.close()
    print(json.dumps({"case": name, "csr_us": round(t_csr * 1e3, 1), "row_band_us": round(t_rb * 1e3, 1), "panel_us": round(t_pb * 1e3, 1)}), flush=True)
    # the split: per nonzero the number of nonzeros of its band in its line of x
    deg_t = (csr.offsets[1:] - csr.offsets[:-1]).long()
    row_of = torch.repeat_interleave(torch.arange(rows, device="cuda"), deg_t)
    lines = (cols + 31) // 32
    key = (row_of // H) * lines + (csr.indices.long() >> 5)
    _, inv, cnt = torch.unique(key, return_inverse=True, return_counts=True)
    per_item = cnt[inv]
    del key, inv, cnt
    for T in Ts:
        dense = per_item >= T
        def sub(mask):
            d = torch.zeros(rows + 1, dtype=torch.long, device="cuda")
            d[1:] = torch.cumsum(torch.bincount(row_of[mask], minlength=rows), 0)
            return S.CSR(rows, cols, d.int().contiguous(), csr.indices[mask].contiguous(), csr.values[mask].contiguous())
        a_d, a_s = sub(dense), sub(~dense)
        share = a_d.nnzs / nnz
        rb = S.RowBandPlan(a_d, H); rb.tune(3)
        pb = S.PanelBinnedPlan(a_s)
        rb.spmv(x, y1); pb.spmv(x, y2)
        eq = bool(torch.equal(y1 + y2, y0))   # (exactly summable inputs)
        t_d = batch_ms(lambda: rb.spmv(x, y1), iters=10)
        t_s = batch_ms(lambda: pb.spmv(x, y2), iters=10)
        t_both = batch_ms(lambda: (rb.spmv(x, y1), pb.spmv(x, y2)), iters=10)
        print(json.dumps({"case": name, "T": T, "dense_share": round(share, 4), "row_band_dense_us": round(t_d * 1e3, 1), "panel_sparse_us": round(t_s * 1e3, 1),
                          "both_us": round(t_both * 1e3, 1), "sum_equals_csr": eq, "row_band_padding_pct": round(100.0 * (rb.padded - a_d.nnzs) / max(a_d.nnzs, 1), 2)}), flush=True)
        rb.close(); pb.close()
        del a_d, a_s, dense
    del csr, x, row_of, per_item
    torch.cuda.empty_cache()
