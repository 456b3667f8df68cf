"""Column-blocked layout vs plain CSR on shapes whose x exceeds the per-XCD L2:
(a) rank shards of an N-GPU weak-scaling run (2^20 rows x N*2^20 cols, 2^24 nnz), (b) a C3-sized
scale-free stand-in (7.4 M rows, 194 M nnz, x = 30 MB).  Exactly-summable inputs: results must be equal."""
import argparse, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from loops_amd import generate as G, spmv as S

def ev(fn, iters=30, warm=3):
    for _ in range(warm): fn()
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(iters)]
    for a, b in evs:
        a.record(); fn(); b.record()
    torch.cuda.synchronize()
    return float(np.median([a.elapsed_time(b) for a, b in evs]))

def gen(rows, cols, nnz):
    deg = G.powerlaw_degrees(rows, nnz)
    parts, bounds = [], np.linspace(0, rows, max(1, nnz >> 25) + 1).astype(np.int64)
    for a, b in zip(bounds[:-1], bounds[1:]):
        parts.append(G.csr_from_degrees(deg[a:b], cols, 1, int(a), True, None))
    off = np.concatenate([[0]] + [p[0][1:].astype(np.int64) + sum(int(q[0][-1]) for q in parts[:k]) for k, p in enumerate(parts)]).astype(np.int32)
    return off, np.concatenate([p[1] for p in parts]), np.concatenate([p[2] for p in parts])

ap = argparse.ArgumentParser(); ap.add_argument("--c3", action="store_true"); a = ap.parse_args()
cases = [(f"shard of N={n}", 1 << 20, n << 20, 1 << 24) for n in (1, 2, 4, 8)]
if a.c3: cases.append(("C3 stand-in", 7414866, 7414866, 194109311))
out = {}
for name, rows, cols, nnz in cases:
    off, idx, val = gen(rows, cols, nnz)
    csr = S.CSR.from_numpy(rows, cols, off, idx, val)
    x = torch.from_numpy(G.uniform_distribution_int(cols)).cuda()
    plan = S.MergePathPlan(csr); y = torch.empty(rows, device="cuda")
    t0 = ev(lambda: S.merge_path_flat(csr, x, y, plan=plan))
    cb = S.ColumnBlockedPlan(csr); y2 = torch.empty(rows, device="cuda")
    t1 = ev(lambda: cb.spmv(x, y2))
    st = [ev(lambda s=s: cb.spmv_stage(s, x, y2)) for s in (0, 1, 2)]
    other = {}
    for sch in ("work_oriented", "group_mapped"):
        tp = ev(lambda: S.spmv(sch, csr, x, y), 10)
        tb = ev(lambda: cb.spmv_schedule(sch, x, y2), 10)
        other[sch] = {"plain_ms": round(tp, 4), "blocked_ms": round(tb, 4), "equal": bool(torch.equal(y, y2))}
    S.merge_path_flat(csr, x, y, plan=plan)
    cb.spmv(x, y2); torch.cuda.synchronize()
    out[name] = {"rows": rows, "cols": cols, "nnz": nnz, "x_MB": cols * 4 >> 20, "plain_ms": round(t0, 4), "blocks": cb.num_blocks,
                 "blocked_ms": round(t1, 4), "stages_ms": [round(s, 4) for s in st], "equal": bool(torch.equal(y, y2)),
                 "GFLOPs_plain": round(2 * nnz / t0 / 1e6, 1), "GFLOPs_blocked": round(2 * nnz / t1 / 1e6, 1),
                 "other_schedules": other}
    print(name, out[name], file=sys.stderr, flush=True)
    del csr, cb, plan
print(json.dumps(out))
