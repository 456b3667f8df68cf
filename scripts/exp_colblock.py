"""Experiment: column-blocked ("stacked") CSR for shards whose x does not fit L2.
A shard of 2^20 rows x (N * 2^20) cols, 2^24 nnz (what rank r of an N-GPU weak-scaling run holds):
plain merge_path_flat vs the same kernel on the stacked CSR (row k*rows + r = columns of block k of
row r) followed by a K-way row reduce."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from loops_amd import generate as G, spmv as S
def ev(fn, iters=30, warm=3):
    for _ in range(warm): fn()
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(iters)]
    for a, b in evs:
        a.record(); fn(); b.record()
    torch.cuda.synchronize()
    return float(np.median([a.elapsed_time(b) for a, b in evs]))
rows, nnz = 1 << 20, 1 << 24
for N in (1, 2, 4, 8):
    cols = N << 20
    deg = G.powerlaw_degrees(rows, nnz)
    off, idx, val = G.csr_from_degrees(deg, cols, 1, 0, True, None)
    off = off.astype(np.int64)
    x = torch.from_numpy(G.uniform_distribution_int(cols)).cuda()
    csr = S.CSR.from_numpy(rows, cols, off, idx, val)
    plan = S.MergePathPlan(csr)
    y = torch.empty(rows, device="cuda")
    t0 = ev(lambda: S.merge_path_flat(csr, x, y, plan=plan))
    line = f"N={N} x={cols*4>>20} MB  plain {t0*1e3:7.1f} us"
    for K in sorted({N, 2 * N, 8} if N > 1 else {2, 8}):
        bw = cols // K
        blk = idx // bw                                   # block of every nonzero
        rowid = np.repeat(np.arange(rows, dtype=np.int64), np.diff(off))
        key = blk.astype(np.int64) * rows + rowid          # stacked row
        order = np.argsort(key, kind="stable")
        soff = np.zeros(K * rows + 1, np.int64)
        np.add.at(soff, key + 1, 1); soff = np.cumsum(soff)
        scsr = S.CSR.from_numpy(K * rows, cols, soff, idx[order], val[order])
        splan = S.MergePathPlan(scsr)
        ys = torch.empty(K * rows, device="cuda")
        def run():
            S.merge_path_flat(scsr, x, ys, plan=splan)
            torch.sum(ys.view(K, rows), dim=0, out=y2)
        y2 = torch.empty(rows, device="cuda")
        t1 = ev(run)
        tk = ev(lambda: S.merge_path_flat(scsr, x, ys, plan=splan))
        assert torch.equal(y, y2)
        line += f" | K={K}: {t1*1e3:7.1f} us (spmv {tk*1e3:.1f})"
    print(line, flush=True)
