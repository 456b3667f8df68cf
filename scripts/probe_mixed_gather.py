"""Scalar + vector gathers: G gathers/s of out[i] = table[idx[i]] (2^24 random indices) when K of every 64 gathers of a
wavefront go through the scalar memory path; tables of 1 MB (always L2 hits) and 4 MB (C2's x)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from loops_amd import probes as PR

def ev(fn, iters=20):
    for _ in range(3): fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / iters

n = 1 << 24
out = torch.empty(n, dtype=torch.float32, device="cuda")
for log in (18, 20):
    table = torch.arange(1 << log, dtype=torch.float32, device="cuda")
    idx = torch.randint(0, 1 << log, (n,), dtype=torch.int32, device="cuda")
    row = []
    for k in (0, 4, 8, 16, 24, 32, 64):
        ms = ev(lambda: PR.mixed_gather(table, idx, out, k))
        ok = bool(torch.equal(out, idx.float()))
        row.append(f"K={k:2d} {n/ms/1e6:6.1f} G/s{'' if ok else ' WRONG'}")
    print(f"table {4 << (log - 20) if log >= 20 else (4 << log) >> 20} MB: " + "  ".join(row), flush=True)
