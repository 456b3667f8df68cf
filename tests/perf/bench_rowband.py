"""Row-band layout (loops_rowband_plan_*): layout check against the numpy specification, equality with the CSR product, and a
sweep of band height x chunk size x kernel shape on one case.  usage: bench_rowband.py [case] [--f64] [--nocheck]
env: RB_H=4096,8192,16384  RB_G=0,256,512 (target chunks)  RB_CFG=162,164,82,84  RB_DIAG=1,2,3  (comma lists)"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import numpy as np, torch
from loops_amd import generate as G, spmv as S
from bench_panel_cases import CASES, batch_ms, build_ms  # noqa: E402
import rowband_spec as spec  # noqa: E402

name = next((a for a in sys.argv[1:] if a in CASES), "c2")
rows, cols, nnz, window = CASES[name]
deg = G.powerlaw_degrees(rows, nnz, cap=min(1 << 14, cols)) if name != "short_rows_8M" else np.full(rows, 2, np.int64)
if "--uniform-degrees" in sys.argv:
    deg = np.full(rows, nnz // rows, np.int64)
hosts = G.host_blocks(cols) if window == G.HOST_BLOCKED else None
off, idx, val = G.csr_from_degrees(deg, cols, 1, 0, True, window, hosts=hosts)
F64 = "--f64" in sys.argv   # 8-byte values: the CSR kernel of the same precision beside the row-band copy
VB = 8 if F64 else 4
if F64:
    val = val.astype(np.float64)
csr = S.CSR.from_numpy(rows, cols, off, idx, val)
x = torch.from_numpy(G.uniform_distribution_int(cols).astype(val.dtype)).cuda()
y0, y1 = torch.empty(rows, dtype=x.dtype, device="cuda"), torch.empty(rows, dtype=x.dtype, device="cuda")
abytes = nnz * (4 + VB) + (rows + 1) * 4 + rows * VB + cols * VB
mp = S.MergePathPlan(csr, "512x8")
S.merge_path_flat(csr, x, y0, plan=mp)
t_csr = batch_ms(lambda: S.merge_path_flat(csr, x, y0, plan=mp))
print(name, "csr 512x8 %.1f us" % (t_csr * 1e3), file=sys.stderr, flush=True)

def ints(key, default):
    return [int(t) for t in os.environ.get(key, default).split(",") if t]

out = {"case": name, "value_bytes": VB, "csr_512x8_us": round(t_csr * 1e3, 2), "csr_frac": round(abytes / t_csr / 1e6 / 8000, 4), "runs": []}
checked = False
for H in ints("RB_H", "8192"):
    plan, b_ms = build_ms(lambda: S.RowBandPlan(csr, H, 0))
    if not checked and "--nocheck" not in sys.argv:   # the device-built layout against the specification (once)
        v, r16, d8, perm, base, bs, hubs = spec.layout(off, idx, val, rows, cols, H)
        dv, dr16, dd8, dperm, dbase, dchunks, dmulti, dhubs = plan.arrays()
        assert plan.steps == base.shape[0], (plan.steps, base.shape)
        assert np.array_equal(dhubs, hubs), (dhubs[:2], hubs[:2])
        assert np.array_equal(dv, v) and np.array_equal(dr16, r16) and np.array_equal(dd8, d8) and np.array_equal(dperm, perm) and np.array_equal(dbase, base)
        ch, mu = spec.chunk_list(bs, plan.num_bands if 2 * plan.num_bands > 256 else 256)
        assert np.array_equal(dchunks[:, :4], ch) and np.array_equal(dmulti, mu), (dchunks[:4], ch[:4])
        checked = True
        print("layout == specification (H %d, steps %d, padding %.2f %%)" % (H, plan.steps, 100.0 * (plan.padded - nnz) / max(nnz, 1)), file=sys.stderr, flush=True)
    for CHK in ints("RB_G", "0"):
        if checked is True and CHK:
            pass
        plan.set_chunks(CHK)
        for cfg in ints("RB_CFG", "162"):
            plan.set_waves(16 if cfg // 10 == 16 else 8)
            plan.spmv(x, y1)
            eq = bool(torch.equal(y0, y1))
            diag = {}
            t = batch_ms(lambda: plan.spmv(x, y1))
            ta = batch_ms(lambda: plan.spmv_stage(0, x, y1))
            tb = batch_ms(lambda: plan.spmv_stage(1, x, y1)) if plan.num_multi else 0.0
            row = {"H": H, "target_chunks": CHK, "cfg": cfg, "chunks": plan.num_chunks, "partials": plan.num_partials,
                   "us": round(t * 1e3, 2), "accumulate_us": round(ta * 1e3, 2), "combine_us": round(tb * 1e3, 2),
                   "frac": round(abytes / t / 1e6 / 8000, 4), "equal": eq, "build_ms": round(b_ms, 2),
                   "padding_pct": round(100.0 * (plan.padded - nnz) / max(nnz, 1), 2), **diag}
            out["runs"].append(row)
            print(json.dumps(row), file=sys.stderr, flush=True)
    plan.close()
print(json.dumps(out))
