"""group_mapped next to work_oriented / merge_path_flat (whole one-shot calls through loops_spmv_csr_f32) on the structures the
round-5 review names: R-MAT 2^23 x 23 in generator order (hub rows first), the host-blocked and band C3 stand-ins, C2.
usage: bench_group_mapped.py [rmat|host|band|c2|c2x8|c2x16|uniform ...]"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from loops_amd import generate as G, spmv as S
from oracle import oracle as O


def per_call_ms(fn, iters=10, warm=2, rounds=3):
    for _ in range(warm): fn()
    t = []
    for _ in range(rounds):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(iters): fn()
        b.record(); torch.cuda.synchronize()
        t.append(a.elapsed_time(b) / iters)
    return float(np.median(t))


def matrix(name):
    if name == "rmat":
        off, idx, val = G.rmat_csr(23, 23, relabel="none")
        return off, idx, val, 1 << 23, 1 << 23
    if name in ("c2", "c2x8", "c2x16"):   # C2, and C2's rows over an x of 8 / 16 MB
        rows = 1 << 20
        cols = rows * {"c2": 1, "c2x8": 2, "c2x16": 4}[name]
        off, idx, val = G.powerlaw_csr(rows, cols, 1 << 24)
        return off, idx, val, rows, cols
    rows = cols = 7_414_866
    deg = G.powerlaw_degrees(rows, 194_109_311, native=True)
    window = {"host": G.HOST_BLOCKED, "band": 65536, "uniform": None}[name]
    hosts = G.host_blocks(cols) if window == G.HOST_BLOCKED else None
    off, idx, val = G.csr_from_degrees(deg, cols, 1, 0, True, window, native=True, hosts=hosts)
    return off, idx, val, rows, cols


out = {}
for name in (sys.argv[1:] or ["rmat", "host", "band", "c2"]):
    off, idx, val, rows, cols = matrix(name)
    x = G.uniform_distribution_int(cols)
    want = O.spmv_f32(off, idx, val, x, omp=True)
    csr = S.CSR.from_numpy(rows, cols, off, idx, val)
    xd = torch.from_numpy(x).cuda(); y = torch.empty(rows, device="cuda")
    row = {}
    for sched in ("group_mapped", "work_oriented", "merge_path_flat"):
        iters = 50 if name.startswith("c2") else 10
        ms = per_call_ms(lambda: S.spmv(sched, csr, xd, y), iters=iters)
        y.fill_(-1.0); S.spmv(sched, csr, xd, y)
        row[sched] = {"ms": round(ms, 4), "bit_exact": bool(np.array_equal(y.cpu().numpy(), want))}
    out[name] = row
    print(name, {k: (v["ms"], v["bit_exact"]) for k, v in row.items()}, file=sys.stderr)
    del csr, xd, y
print(json.dumps(out))
