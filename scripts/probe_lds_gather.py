"""LDS-direct gathers (global_load_lds_dword) vs ordinary gathers: G gathers/s of out[i] = table[idx[i]], 2^24 indices."""
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
for log in (18, 20, 23):
    table = torch.arange(1 << log, dtype=torch.float32, device="cuda")
    idx = torch.randint(0, 1 << log, (n,), dtype=torch.int32, device="cuda")
    row = []
    for mode, name in ((0, "VGPR"), (5, "LDS x4"), (6, "LDS x8"), (7, "LDS x16")):
        out.zero_()
        ms = ev(lambda: PR.gather(table, idx, out, mode))
        frac_ok = float((out == idx.float()).float().mean())
        row.append(f"{name} {n/ms/1e6:6.1f} G/s (ok {frac_ok:.3f})")
    print(f"table {(4 << log) >> 20} MB: " + "  ".join(row), flush=True)
