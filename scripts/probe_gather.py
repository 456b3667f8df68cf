"""Microbenchmark: random 4-byte gather rate vs table size and load flavour (calibrates what
bounds the x[col] reads of SpMV on this box)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from loops_amd import probes as S

def ev(fn, iters=20):
    for _ in range(3): fn()
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(iters)]
    for a, b in evs:
        a.record(); fn(); b.record()
    torch.cuda.synchronize()
    return float(np.median([a.elapsed_time(b) for a, b in evs]))

n = 1 << 24
g = torch.Generator(device="cuda"); g.manual_seed(1)
out = torch.empty(n, dtype=torch.float32, device="cuda")
for logt in (14, 16, 18, 20, 22, 24, 26):
    t = 1 << logt
    table = torch.rand(t, device="cuda")
    idx = torch.randint(0, t, (n,), device="cuda", dtype=torch.int32, generator=g)
    sidx = torch.sort(idx.view(-1, 64), dim=1).values.view(-1).contiguous()  # sorted within a wavefront's 64
    line = (torch.arange(n, device="cuda", dtype=torch.int32) % t)             # perfectly coalesced
    row = []
    for mode in (0, 1, 2, 3, 4):
        ms = ev(lambda: S.gather(table, idx, out, mode))
        row.append(f"m{mode} {n/ms/1e6:7.1f}")
    ms = ev(lambda: S.gather(table, sidx, out, 0)); row.append(f"sorted64 {n/ms/1e6:7.1f}")
    ms = ev(lambda: S.gather(table, line, out, 0)); row.append(f"coalesced {n/ms/1e6:7.1f}")
    print(f"table 2^{logt} floats ({t*4/2**20:8.2f} MiB): Gelem/s " + "  ".join(row), flush=True)
# streaming copy at several sizes
for logn in (22, 24, 25, 26, 27, 28):
    n2 = 1 << logn
    src = torch.rand(n2, device="cuda"); dst = torch.empty_like(src)
    ms = ev(lambda: S.stream_copy(src, dst))
    ms2 = ev(lambda: dst.copy_(src))
    ms3 = ev(lambda: S.stream_copy(src, src))
    print(f"copy 2^{logn} floats: ours {2*n2*4/ms/1e6:7.1f} GB/s  torch {2*n2*4/ms2/1e6:7.1f} GB/s  read-only {n2*4/ms3/1e6:7.1f} GB/s", flush=True)
